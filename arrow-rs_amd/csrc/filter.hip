// filter.hip — arrow_select::filter on MI355X.
//
// Reference path (arrow-select/src/filter.rs): filter :201 -> FilterBuilder::new
// :256 (true_count, prep_null_mask_filter :167) -> filter_array :535 ->
// filter_primitive :773 { filter_native :731, filter_nulls :512 { filter_bits :680,
// count_set_bits } }.  The reference makes 4 passes over the mask (count, value
// gather, bit gather, null count) with a tzcnt loop per set bit
// (arrow-buffer/src/util/bit_iterator.rs:284-324).
//
// MI355X design (not a translation):
//   K1 filter_count_kernel   one popcount pass over (mask & mask-validity); the
//                            mask word IS the ballot.  Per-1024-row "chunk"
//                            exclusive prefixes + per-group totals, one block
//                            per 1024 chunks, LDS block scan — no atomics.
//   K2 filter_group_scan     scans <=1024 group totals per pass -> u64 group
//                            prefixes + K (the only number the host waits for).
//   K3 filter_scatter_kernel one tile (<=4096 rows) per workgroup.  16-byte
//                            coalesced value loads are issued first; wave 0
//                            meanwhile builds the tile's word table in LDS
//                            (mask, validity, exclusive popcount prefix).
//                            rank(row) = prefix[word] + popc(word & below(row)).
//                            Selected values and their validity flags are
//                            compacted through a 16 KiB LDS stage (2048 rows
//                            per chunk, so 8 workgroups stay resident per CU;
//                            a tile with more selected rows loops over chunks
//                            while the values stay in registers) and leave
//                            with coalesced stores; validity bits are
//                            re-packed with __ballot aligned to the global
//                            64-bit output words (interior words plain-stored,
//                            boundary words merged with a 64-bit atomicOr).
//                            Below ~12% selectivity the 16-byte loads are
//                            predicated on "any row of this lane selected":
//                            128-byte lines without a selected row are never
//                            fetched (18.5% of them at 10% selectivity).
//                            Valid-row counts go to 64 atomic slots (one add
//                            per tile) -> null_count without a reduce pass.
// Values are read at most once; the mask twice (1.4% of the bytes at Int64).
#include "common.hpp"
#include "filter_internal.hpp"
#include "window_tiles.hpp"

namespace {

constexpr int CHUNK_ROWS = 1024;            // count-pass granule (16 mask words)
constexpr int GROUP_CHUNKS = 1024;          // chunks per count block
constexpr int SCATTER_THREADS = 256;

// (tile_rows: filter_internal.hpp)

// ------------------------------------------------------------------ K1
__global__ void __launch_bounds__(1024) filter_count_kernel(BitView mask, BitView mask_valid,
                                                            int64_t len, uint32_t* chunk_prefix,
                                                            uint32_t* group_total) {
  __shared__ uint32_t s_cnt[GROUP_CHUNKS];
  __shared__ uint32_t s_wave[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t chunk_base = (int64_t)blockIdx.x * GROUP_CHUNKS;
  // four steps' words (mask and its validity) are requested before the first one is counted: bv_fetch64 in the loop
  // waits for each word in turn (16 serialized round trips per wave)
  const bool has_mv = mask_valid.words != nullptr;
  for (int it0 = 0; it0 < 16; it0 += 4) {
    BvRaw rm[4], rv[4] = {};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t s = ((chunk_base + wave * 64 + (it0 + j) * 4) * 16 + lane) << 6;  // 4 chunks = 64 words per wave step
      rm[j] = bv_issue(mask, s < len ? s : 0, len);
    }
    if (has_mv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t s = ((chunk_base + wave * 64 + (it0 + j) * 4) * 16 + lane) << 6;
        rv[j] = bv_issue(mask_valid, s < len ? s : 0, len);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j;
      const int64_t s = ((chunk_base + wave * 64 + it * 4) * 16 + lane) << 6;
      const int64_t sc = s < len ? s : 0;
      uint64_t m = bv_finish(rm[j], sc, len);
      if (has_mv) m &= bv_finish(rv[j], sc, len);
      if (s >= len) m = 0;
      int c = __popcll(m);
      c += __shfl_xor(c, 1, 64);
      c += __shfl_xor(c, 2, 64);
      c += __shfl_xor(c, 4, 64);
      c += __shfl_xor(c, 8, 64);
      if ((lane & 15) == 0) s_cnt[wave * 64 + it * 4 + (lane >> 4)] = (uint32_t)c;
    }
  }
  __syncthreads();
  int v = (int)s_cnt[t];
  int incl = wave_scan_incl(v);
  if (lane == 63) s_wave[wave] = (uint32_t)incl;
  __syncthreads();
  uint32_t wbase = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) wbase += (w < wave) ? s_wave[w] : 0u;
  int64_t nchunks = (len + CHUNK_ROWS - 1) / CHUNK_ROWS;
  if (chunk_base + t < nchunks) chunk_prefix[chunk_base + t] = wbase + (uint32_t)(incl - v);
  if (t == 1023) group_total[blockIdx.x] = wbase + (uint32_t)incl;
}

// K1 for small predicates (<= 64 Mi rows: BatchCoalescer / DataFusion-sized batches).  The kernel above gives one
// 1024-thread block per 1 Mi rows — 16 blocks on 256 CUs for a 2^24-row batch (21 us).  Here a block is ONE wave that
// owns 64 chunks (65 536 rows), so the same batch spreads over 256 blocks; the group granule becomes 64 chunks
// (`group_shift` 6 instead of 10) and the scatter reads its prefix with that shift.
// (Folding K2 in was tried twice and dropped: with release / acquire fences in round 2 — 16.7 us against 6.5 + 3.9 us for
// the two launches — and in round 3 fence-free, every block adding {1 arrival, its count} to one 64-bit ticket whose last
// arriver posts K, the scatter summing the group totals itself: the count went from 6.4 to 11.8 us (256 same-address
// atomics + the system-scope post) and every scatter tile paid an extra dependent L2 round trip, 49 -> 54 us per launch.)
__global__ void __launch_bounds__(64) filter_count_small_kernel(BitView mask, BitView mask_valid, int64_t len,
                                                               uint32_t* chunk_prefix, uint32_t* group_total) {
  __shared__ uint32_t s_cnt[64];
  const int lane = threadIdx.x;
  const int64_t chunk_base = (int64_t)blockIdx.x * 64;
  // every word of the wave's 64 chunks is requested before the first one is counted (bv_fetch64 in the loop made
  // this 16 serialized memory round trips per wave: 17-20 us for a 2 MiB mask, 6.5 us now).  Folding K2 into this
  // kernel ("last wave done" ticket) was measured and dropped: the agent-scope release / acquire fences it needs
  // write back and invalidate L2 on a multi-XCD part, 16.7 us against 6.5 + 3.9 us for the two launches.
  const bool has_mv = mask_valid.words != nullptr;
  BvRaw rm[16], rv[16] = {};
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;  // 4 chunks = 64 words per step, coalesced
    rm[it] = bv_issue(mask, s < len ? s : 0, len);
  }
  if (has_mv) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
      rv[it] = bv_issue(mask_valid, s < len ? s : 0, len);
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
    const int64_t sc = s < len ? s : 0;
    uint64_t m = bv_finish(rm[it], sc, len);
    if (has_mv) m &= bv_finish(rv[it], sc, len);
    if (s >= len) m = 0;
    int c = __popcll(m);
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 8, 64);
    if ((lane & 15) == 0) s_cnt[it * 4 + (lane >> 4)] = (uint32_t)c;
  }
  __syncthreads();
  const int v = (int)s_cnt[lane];
  const int incl = wave_scan_incl(v);
  const int64_t nchunks = (len + CHUNK_ROWS - 1) / CHUNK_ROWS;
  if (chunk_base + lane < nchunks) chunk_prefix[chunk_base + lane] = (uint32_t)(incl - v);
  if (lane == 63) group_total[blockIdx.x] = (uint32_t)incl;
}

// K1 for up to 8 small predicates in ONE launch (blockIdx.y = predicate): the grouped push of BatchCoalescer counted 60
// batches per 1e9 rows with 120 launches of ~6 us each — 0.7 ms of a 4 ms step.
struct CountMulti {
  struct One {
    BitView mask, mask_valid;
    int64_t len;
    uint32_t* chunk_prefix;
    uint32_t* group_total;
    unsigned long long* group_prefix;
    unsigned long long* total;
    int64_t ngroups;
    uint64_t* quant_pin;  // device view of pinned words (or null): prefix of every quant_step-th group, for the host
    int64_t quant_step;
  } p[8];
};
__global__ void __launch_bounds__(64) filter_count_small_multi_kernel(CountMulti m) {
  const CountMulti::One& o = m.p[blockIdx.y];
  if ((int64_t)blockIdx.x >= o.ngroups) return;
  __shared__ uint32_t s_cnt[64];
  const int lane = threadIdx.x;
  const int64_t chunk_base = (int64_t)blockIdx.x * 64, len = o.len;
  const bool has_mv = o.mask_valid.words != nullptr;
  BvRaw rm[16], rv[16] = {};
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
    rm[it] = bv_issue(o.mask, s < len ? s : 0, len);
  }
  if (has_mv) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
      rv[it] = bv_issue(o.mask_valid, s < len ? s : 0, len);
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
    const int64_t sc = s < len ? s : 0;
    uint64_t mk = bv_finish(rm[it], sc, len);
    if (has_mv) mk &= bv_finish(rv[it], sc, len);
    if (s >= len) mk = 0;
    int c = __popcll(mk);
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 8, 64);
    if ((lane & 15) == 0) s_cnt[it * 4 + (lane >> 4)] = (uint32_t)c;
  }
  __syncthreads();
  const int v = (int)s_cnt[lane];
  const int incl = wave_scan_incl(v);
  const int64_t nchunks = (len + CHUNK_ROWS - 1) / CHUNK_ROWS;
  if (chunk_base + lane < nchunks) o.chunk_prefix[chunk_base + lane] = (uint32_t)(incl - v);
  if (lane == 63) o.group_total[blockIdx.x] = (uint32_t)incl;
}
// K2 of the same: block b scans predicate b's group totals (<= 1024 of them) and sends its K to pinned slot slot0 + b
__global__ void __launch_bounds__(1024) filter_group_scan_multi_kernel(CountMulti m, uint64_t* mail, int slot0) {
  const CountMulti::One& o = m.p[blockIdx.x];
  __shared__ unsigned long long s_wave[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long v = t < o.ngroups ? o.group_total[t] : 0ull, incl = v;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    unsigned long long u = __shfl_up(incl, k, 64);
    if (lane >= k) incl += u;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += s_wave[w];
  if (t < o.ngroups) {
    o.group_prefix[t] = wbase + incl - v;
    if (o.quant_pin && t % o.quant_step == 0)
      __hip_atomic_store(o.quant_pin + t / o.quant_step, (uint64_t)(wbase + incl - v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t == 1023) {
    *o.total = wbase + incl;
    __hip_atomic_store(mail + slot0 + blockIdx.x, (uint64_t)(wbase + incl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------ K2
__global__ void __launch_bounds__(1024) filter_group_scan_kernel(const uint32_t* group_total,
                                                                 int64_t ngroups,
                                                                 unsigned long long* group_prefix,
                                                                 unsigned long long* total_out, uint64_t* mail, int slot,
                                                                 uint64_t seq) {
  __shared__ unsigned long long s_wave[16];
  __shared__ unsigned long long s_carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < ngroups; base += 1024) {
    unsigned long long v = (base + t < ngroups) ? group_total[base + t] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wbase = s_carry;
    for (int w = 0; w < wave; ++w) wbase += s_wave[w];
    if (base + t < ngroups) group_prefix[base + t] = wbase + incl - v;
    __syncthreads();
    if (t == 1023) s_carry = wbase + incl;
    __syncthreads();
  }
  if (t == 0) {
    *total_out = s_carry;
    if (mail) {  // K goes straight to the host's pinned slot `slot`: the only number the host waits for
      __hip_atomic_store(mail + slot, (uint64_t)s_carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (seq) ah_mail_post(mail, seq);  // on the mailbox BASE (the sequence word is mail[AH_MAIL_FLAG]); seq == 0: one of
                                         // several counts, a later kernel posts for all of them
    }
  }
}

// the no-wait tail of ah_filter_predicate_apply_into_acc: *acc += K - (valid rows), counters back to zero
__global__ void __launch_bounds__(64) filter_finish_acc_kernel(unsigned long long* slots, unsigned long long k,
                                                               unsigned long long* acc) {
  unsigned long long v = slots[threadIdx.x];
  slots[threadIdx.x] = 0;
  v = wave_reduce_add64(v);
  if (threadIdx.x == 0 && acc && k != v) atomicAdd(acc, k - v);
}

// the same for the columns of one fused launch: block c folds column c's counters
__global__ void __launch_bounds__(64) filter_finish_acc_cols_kernel(unsigned long long* slots, unsigned long long k,
                                                                    unsigned long long* acc) {
  unsigned long long* my = slots + (size_t)blockIdx.x * 64;
  unsigned long long v = my[threadIdx.x];
  my[threadIdx.x] = 0;
  v = wave_reduce_add64(v);
  if (threadIdx.x == 0 && k != v) atomicAdd(acc + blockIdx.x, k - v);
}

// the output bitmaps of a fused multi-column scatter, zeroed by one launch (blockIdx.y = column)
struct ZeroCols { unsigned long long* p[8]; };
__global__ void __launch_bounds__(256) zero_words_cols_kernel(ZeroCols z, int64_t words) {
  unsigned long long* p = z.p[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

// the speculative launch's tail: K is still on its way to the host, so the window's row count comes from the device
__global__ void __launch_bounds__(64) filter_finish_acc_cols_dev_kernel(unsigned long long* slots, const unsigned long long* total,
                                                                        unsigned long long win_lo, unsigned long long win_hi,
                                                                        unsigned long long* acc) {
  unsigned long long* my = slots + (size_t)blockIdx.x * 64;
  unsigned long long v = my[threadIdx.x];
  my[threadIdx.x] = 0;
  v = wave_reduce_add64(v);
  const unsigned long long hi = *total < win_hi ? *total : win_hi;
  const unsigned long long k = hi > win_lo ? hi - win_lo : 0;
  if (threadIdx.x == 0 && k != v) atomicAdd(acc + blockIdx.x, k - v);
}

// one wait for a fused multi-column scatter: column c's valid-row count lands in mail[c], counters back to zero
__global__ void __launch_bounds__(64) filter_finish_cols_kernel(unsigned long long* slots, int ncols, uint64_t* mail,
                                                                uint64_t seq) {
  for (int c = 0; c < ncols; ++c) {
    unsigned long long v = slots[c * 64 + threadIdx.x];
    slots[c * 64 + threadIdx.x] = 0;
    v = wave_reduce_add64(v);
    if (threadIdx.x == 0) __hip_atomic_store(mail + c, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) ah_mail_post(mail, seq);
}

// after the scatter: fold the VALID_SLOTS counters into one number for the host, leave them zero for the next
// call (ctx->scratch is self-cleaning: no per-call memset), publish.  mail == nullptr (deferred): clean only.
__global__ void __launch_bounds__(64) filter_finish_kernel(unsigned long long* slots, uint64_t* mail, uint64_t seq) {
  unsigned long long v = slots[threadIdx.x];
  slots[threadIdx.x] = 0;
  v = wave_reduce_add64(v);
  if (threadIdx.x == 0 && mail) {
    __hip_atomic_store(mail, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ah_mail_post(mail, seq);
  }
}

// ------------------------------------------------------------------ K3
// (Elem / Vec: filter_internal.hpp)

struct ScatterArgs {
  const void* values;
  BitView mask, mask_valid, vvalid;
  int64_t len;  // predicate length
  const uint32_t* chunk_prefix;
  const unsigned long long* group_prefix;
  int nulls_mode;  // 1: the slots accumulate NULL rows appended (window rows - valid rows) instead of valid rows
  void* out_values;
  unsigned long long* out_valid;   // zero-initialised u64 words
  unsigned long long* valid_slots; // VALID_SLOTS zero-initialised counters (valid selected rows)
  int group_shift;                 // log2(chunks per count group): 10 (filter_count_kernel) or 6 (the small-mask one)
  int64_t chunk_base;              // batch tables (ah_tbl_*): the predicate's first chunk in the PUSH's chunk numbering — the prefix
                                   // arrays are indexed by global chunk there (a count wave spans batches); 0 everywhere else
  int xcd_remap;                   // 1: tiles that share output lines stay on one XCD
  int64_t out_base;                // rows already present in the destination (fused filter-into-builder)
  int64_t ntiles;
  // window of the filtered stream this launch writes: positions [win_lo, win_hi) land at out_base + (pos - win_lo);
  // win_hi == 0: everything.  BatchCoalescer splits a batch that straddles two output batches into two launches.
  int64_t win_lo, win_hi;
  // further columns of the SAME width / validity shape filtered by the same launch (blockIdx.y = column): a record
  // batch's columns share the predicate, its prefix tables and the destination row offset
  struct Col {
    const void* values;
    BitView vvalid;
    void* out_values;
    unsigned long long* out_valid;
    unsigned long long* valid_slots;
  } more[7];
};
constexpr int SCATTER_MAX_COLS = 8;

constexpr int VALID_SLOTS = 64;

// (stage_cap: filter_internal.hpp)

// WIDTH == 0: bit-only variant (Boolean values / validity-only): compacts the
// `vvalid` stream; no value loads.
// SKIP: predicate each 16-byte load on "any of its rows selected" so cache lines
// with no selected row are never fetched (pays below ~30% selectivity).
// 86 VGPRs = 5 workgroups per CU.  Forcing 6 or 8 through __launch_bounds__(256, w) spills the tile's values to scratch and
// is slower (measured r02: 1.48 ms -> 1.85 ms at w = 6, 3.28 ms at w = 8): the registers ARE the tile.
// One tile of one column: the body shared by the single-batch kernel and the multi-batch kernel (BatchCoalescer's
// grouped push: several input batches scattered into one output window by ONE launch).
// S: 4096-row sub-tiles per workgroup (plain single-batch launches of 1-, 2- and 4-byte columns only).  A 4096-row tile of
// Int8 is 4 KiB of values: eight resident workgroups per CU keep 32 KiB in flight per CU, a quarter of what the latency x
// bandwidth product needs, and every tile pays the same word-table / barrier / prefix-lookup chain as a 32 KiB Int64 tile
// (round 5: filter Int8 ran at 0.31 of peak, Int16 at 0.47, Int64 at 0.86).  With S sub-tiles the workgroup owns S * 4096
// rows: waves 0..S-1 each build one sub-tile's 64-word table, `woff` carries the sub-tile bases.
template <int W, int V, bool HAS_VALID, bool SKIP, int S = 1>
__device__ __forceinline__ void scatter_tile(const ScatterArgs& a, int64_t tile, const void* c_values, BitView c_vvalid,
                                             void* c_out_values, unsigned long long* c_out_valid,
                                             unsigned long long* c_valid_slots) {
  constexpr int WE = W == 0 ? 1 : W;
  constexpr int T = tile_rows(WE) * S;
  constexpr int CAP = stage_cap(WE);
  constexpr int RPT = T / SCATTER_THREADS;
  constexpr int L = RPT / V;
  constexpr int NW = T / 64;  // mask words per tile (<= 64 per sub-tile)
  static_assert(S >= 1 && S <= 4 && (S == 1 || tile_rows(WE) == 4096), "sub-tiles are 4096 rows, one table-building wave each");
  constexpr uint32_t VMASK = (V >= 32) ? 0xFFFFFFFFu : ((1u << V) - 1u);
  using ET = typename Elem<WE>::type;

  __shared__ uint64_t s_m[NW];
  __shared__ uint64_t s_v[HAS_VALID ? NW : 1];
  __shared__ uint32_t s_base[NW];
  __shared__ uint32_t s_wsum[S];  // selected rows per sub-tile
  __shared__ uint32_t s_vc[4];
  constexpr int EPV = (W == 0 || W >= 16) ? 1 : 16 / W;  // elements per 16-byte store
  // STAGED (1- and 2-byte values, S > 1): the tile's values are parked in LDS as loaded (16-byte writes) and the compaction
  // walks the SET BITS of each thread's 32 / 64 mask bits, reading the selected values back by row.  The per-row form
  // below visits every row in a lane slot (~14 vector instructions per row whether or not it is selected, because some
  // lane of the wave selects element e of its vector for every e): at 16 rows per 16-byte load that made filter Int8
  // VALU-bound at 0.31 of the HBM peak.  Here a wave's trip count is the largest popcount among its lanes.
  constexpr bool STAGED = (W == 1 || W == 2) && S > 1;
  __shared__ __attribute__((aligned(16))) ET s_raw[STAGED ? T : 1];
  __shared__ __attribute__((aligned(16))) ET s_vals[W == 0 ? 1 : CAP + EPV];
  __shared__ uint8_t s_flag[HAS_VALID && !STAGED ? CAP : 1];
  // STAGED: the validity of the staged rows is a BITMAP in LDS laid out like the output words (bit `lead + position`), OR-ed
  // in by run — a thread's selected rows are consecutive positions, so their validity bits are one shifted register (round 6;
  // a byte per row written in the walk and read back through a ballot per 64 positions cost a third of the walk's LDS traffic)
  constexpr int NBITS32 = (CAP + 128) / 32;
  __shared__ uint32_t s_bits[STAGED && HAS_VALID ? NBITS32 : 1];
  typedef uint32_t __attribute__((aligned(1))) u32u;  // unaligned LDS dword stores (ds_write_b32 at any byte address on gfx950)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = tile * T;

  // 1. value loads (16 B per lane per load on the aligned path) go out first
  Vec<WE, V> regs[STAGED ? 1 : L];
  E16 raw[STAGED ? L : 1];  // STAGED: the 16 bytes as ONE value (as a Vec of 16 bytes the compiler unpacked every load into 16
                            // byte registers behind a vmcnt(0) and re-packed them for the LDS write: ~25 instructions per load)
  if constexpr (STAGED) {
    static_assert(!SKIP && V * WE == 16, "the staged form loads whole 16-byte vectors unconditionally");
    const ET* vp = (const ET*)c_values;
    // every load is issued, unconditionally: vectors past the end of the column re-read its last vector (their mask bits
    // are 0).  A load under `if (r < len)` into an array element cost a branch, a vmcnt(0) and a register shuffle per load.
    const int64_t last_vec = ((a.len - 1) / V) * V;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      int64_t r = row0 + (int64_t)(l * SCATTER_THREADS + t) * V;
      r = r < last_vec ? r : last_vec;
      raw[l] = ah_ld_stream<ah_nt_l(false)>((const E16*)(vp + r));
    }
  } else if constexpr (W != 0 && !SKIP) {
    const ET* vp = (const ET*)c_values;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      int64_t r = row0 + (int64_t)(l * SCATTER_THREADS + t) * V;
      if (r < a.len) regs[l] = ah_ld_stream<ah_nt_l(false)>((const Vec<WE, V>*)(vp + r));
    }
  }

  // 2. wave w < S builds sub-tile w's word table: mask, validity, exclusive popcount prefix
  if (wave < S) {
    uint64_t m = 0, v = 0;
    const int wi = wave * 64 + lane;
    const int64_t s = row0 + ((int64_t)wi << 6);
    const bool has_mv = a.mask_valid.words != nullptr;
    const bool has_vv = HAS_VALID && c_vvalid.words != nullptr;
    // A tile wholly inside the predicate whose bitmaps all start on a word boundary (every unsliced array) reads its words
    // directly — tile-uniform test, three independent 8-byte loads, no funnel shifts or length masks (~60 of the ~110
    // vector instructions of a table build; round 5)
    const bool word_aligned = row0 + T <= a.len &&
                              ((a.mask.off | (has_mv ? a.mask_valid.off : 0) | (has_vv ? c_vvalid.off : 0)) & 63) == 0;
    if (word_aligned) {
      if (wi < NW) {
        const int64_t wq = s >> 6;
        const uint64_t* pm = a.mask.words + (a.mask.off >> 6);
        const uint64_t* pmv = has_mv ? a.mask_valid.words + (a.mask_valid.off >> 6) : pm;  // (absent: the mask's word again, ignored)
        const uint64_t* pvv = has_vv ? c_vvalid.words + (c_vvalid.off >> 6) : pm;
        const uint64_t xm = pm[wq], xmv = pmv[wq], xvv = pvv[wq];
        m = has_mv ? (xm & xmv) : xm;
        if constexpr (HAS_VALID) v = has_vv ? xvv : ~0ull;
      }
    } else if (wi < NW && s < a.len) {
      // all three bitmaps' words are requested before the first one is used: one memory round trip, not three
      // (an absent bitmap re-reads the mask's words — same lines, no branch between the loads — and is ignored)
      const BvRaw rm = bv_issue(a.mask, s, a.len);
      const BvRaw rmv = bv_issue(has_mv ? a.mask_valid : a.mask, s, a.len);
      const BvRaw rvv = bv_issue(has_vv ? c_vvalid : a.mask, s, a.len);
      m = bv_finish(rm, s, a.len);
      const uint64_t mv = bv_finish(rmv, s, a.len), vv = bv_finish(rvv, s, a.len);
      if (has_mv) m &= mv;
      if constexpr (HAS_VALID) v = has_vv ? vv : bv_fetch64(c_vvalid, s, a.len);
    }
    int c = __popcll(m);
    int incl = wave_scan_incl(c);
    if (wi < NW) {
      s_m[wi] = m;
      if constexpr (HAS_VALID) s_v[wi] = v;
      s_base[wi] = (uint32_t)(incl - c);
    }
    if (lane == 63) s_wsum[wave] = (uint32_t)incl;
  }
  if constexpr (STAGED) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int r0 = (l * SCATTER_THREADS + t) * V;
      *(E16*)(s_raw + r0) = raw[l];
    }
    if constexpr (HAS_VALID)
      if (t < NBITS32) s_bits[t] = 0;
    static_assert(NBITS32 <= SCATTER_THREADS, "one thread per bitmap word");
  }
  __syncthreads();

  uint32_t woff[S];  // selected rows of the tile before sub-tile s
  woff[0] = 0;
#pragma unroll
  for (int q = 1; q < S; ++q) woff[q] = woff[q - 1] + s_wsum[q - 1];
  const int total = (int)(woff[S - 1] + s_wsum[S - 1]);
  if (total == 0) return;
  // (requesting these two words up front, with the values and the bitmap words, was measured in round 5: the Int64 scatter
  // went from 1.31 to 1.35-1.37 ms per 1e9 rows on one box, the narrow forms did not move)
  const int64_t chunk0 = a.chunk_base + row0 / CHUNK_ROWS;
  const int64_t rel = (int64_t)a.group_prefix[chunk0 >> a.group_shift] + a.chunk_prefix[chunk0];  // tile's first output position
  int lo_t = 0, hi_t = total;  // the tile's positions inside the launch's window
  if (a.win_hi != 0) {
    const int64_t lo = a.win_lo - rel, hi = a.win_hi - rel;
    lo_t = lo < 0 ? 0 : (lo > total ? total : (int)lo);
    hi_t = hi < 0 ? 0 : (hi > total ? total : (int)hi);
    if (lo_t >= hi_t) return;
  }
  const int64_t ob = a.out_base + rel - a.win_lo;

  if constexpr (W != 0 && SKIP) {
    const ET* vp = (const ET*)c_values;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      int r0 = (l * SCATTER_THREADS + t) * V;
      uint32_t bits = (uint32_t)(s_m[r0 >> 6] >> (r0 & 63)) & VMASK;
      if (bits) regs[l] = ah_ld_stream<ah_nt_l(false)>((const Vec<WE, V>*)(vp + row0 + r0));
    }
  }

  int vc = 0;

  for (int p0 = lo_t; p0 < hi_t; p0 += CAP) {
    const int phase = (int)((ob + p0) & (EPV - 1));
    const int cnt = (hi_t - p0) < CAP ? (hi_t - p0) : CAP;
    // 3. compact the selected rows whose output position falls in [p0, p0+CAP) into LDS
    [[maybe_unused]] const int lead = (int)((ob + p0) & 63);  // bit of output position p0 inside its validity word
    if constexpr (STAGED) {
      const bool whole = lo_t == 0 && hi_t == total && total <= CAP;
      constexpr int PER = 4 / WE;  // values per dword
      constexpr int RS = T / SCATTER_THREADS;  // consecutive rows per thread: half a mask word or a whole one
      static_assert(RS == 32 || RS == 64, "a thread walks the set bits of 32 or 64 mask bits");
      const int r0 = t * RS, w = r0 >> 6, sh = r0 & 63;
      const uint64_t word = s_m[w];
      uint32_t wsel = woff[0];
#pragma unroll
      for (int q = 1; q < S; ++q) wsel = (r0 >> 12) == q ? woff[q] : wsel;
      uint32_t pos = s_base[w] + wsel + (uint32_t)__popcll(word & ((1ull << sh) - 1ull)) - (uint32_t)p0;
      uint64_t vword = 0;
      if constexpr (HAS_VALID) vword = s_v[w];
#pragma unroll
      for (int h = 0; h < RS / 32; ++h) {  // 32 bits at a time: the walk's arithmetic stays 32-bit
        uint32_t b = (uint32_t)(word >> (sh + 32 * h));
        const uint32_t vb = (uint32_t)(vword >> (sh + 32 * h));
        const ET* src = s_raw + r0 + 32 * h;
        if (whole) {  // (tile-uniform) every selected row of the tile is staged by this one pass: no range test in the walk
          // the run's values leave as whole dwords (PER values each, unaligned LDS stores), its validity bits as one register
          uint32_t acc = 0, vacc = 0, wp = pos + (uint32_t)phase;
          int k = 0, nsel = 0;
          const uint32_t pos_h = pos;
          while (b) {
            const int e = __builtin_ctz(b);
            b &= b - 1;
            acc |= (uint32_t)src[e] << (k * 8 * WE);
            if constexpr (HAS_VALID) vacc |= ((vb >> e) & 1u) << nsel;
            ++nsel;
            if (++k == PER) {
              *(u32u*)(s_vals + wp) = acc;
              wp += PER;
              acc = 0;
              k = 0;
            }
          }
#pragma unroll
          for (int j = 0; j < PER - 1; ++j)
            if (j < k) s_vals[wp + j] = (ET)(acc >> (j * 8 * WE));
          pos += (uint32_t)nsel;
          if constexpr (HAS_VALID) {
            if (vacc) {
              const uint32_t o = (uint32_t)lead + pos_h, sh2 = o & 31u;
              atomicOr(&s_bits[o >> 5], vacc << sh2);
              const uint32_t hi2 = sh2 ? (vacc >> (32u - sh2)) : 0u;
              if (hi2) atomicOr(&s_bits[(o >> 5) + 1], hi2);
            }
          }
        } else {
          while (b) {
            const int e = __builtin_ctz(b);
            b &= b - 1;
            if (pos < (uint32_t)cnt) {  // unsigned: also rejects positions before p0
              s_vals[pos + phase] = src[e];
              if constexpr (HAS_VALID) {
                const uint32_t o = (uint32_t)lead + pos;
                if ((vb >> e) & 1u) atomicOr(&s_bits[o >> 5], 1u << (o & 31u));
              }
            }
            ++pos;
          }
        }
      }
    } else {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      int r0 = (l * SCATTER_THREADS + t) * V;
      int w = r0 >> 6, sh = r0 & 63;
      uint64_t word = s_m[w];
      uint32_t bits = (uint32_t)(word >> sh) & VMASK;
      if (bits) {
        // (a thread's V rows of step l lie in sub-tile l * 256 * V / 4096: a compile-time index into woff)
        uint32_t base = s_base[w] + woff[(l * SCATTER_THREADS * V) >> 12] + (uint32_t)__popcll(word & ((1ull << sh) - 1ull)) - (uint32_t)p0;
        uint32_t vb = 0;
        if constexpr (HAS_VALID) vb = (uint32_t)(s_v[w] >> sh);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if ((bits >> e) & 1u) {
            uint32_t pos = base + (uint32_t)__popc(bits & ((1u << e) - 1u));
            if (pos < (uint32_t)cnt) {  // unsigned: also rejects positions before p0
              if constexpr (W != 0) s_vals[pos + phase] = regs[l].u.e[e];
              if constexpr (HAS_VALID) s_flag[pos] = (uint8_t)((vb >> e) & 1u);
            }
          }
        }
      }
    }
    }
    __syncthreads();

    // 4. coalesced write-out of this chunk
    const int64_t cb = ob + p0;
    if constexpr (W != 0) {
      ET* op = (ET*)c_out_values + cb;
      if constexpr (EPV > 1) {
        // 16-byte stores: LDS slot = output position + phase, so LDS vectors and global
        // vectors share their 16-byte alignment; ragged head/tail go element-wise
        const int first = phase, last = phase + cnt;          // LDS element range [first, last)
        const int vfirst = (first + EPV - 1) / EPV, vlast = last / EPV;  // whole vectors
        ET* gbase = op - phase;                               // 16-byte aligned
        if (vfirst < vlast) {
          for (int j = vfirst + t; j < vlast; j += SCATTER_THREADS)
            ah_st_stream<ah_nt_s(true)>((Vec<WE, EPV>*)(gbase + j * EPV), *(const Vec<WE, EPV>*)(s_vals + j * EPV));
          if (t < vfirst * EPV - first) gbase[first + t] = s_vals[first + t];
          if (t < last - vlast * EPV) gbase[vlast * EPV + t] = s_vals[vlast * EPV + t];
        } else {
          for (int j = first + t; j < last; j += SCATTER_THREADS) gbase[j] = s_vals[j];
        }
      } else {
        for (int j = t; j < cnt; j += SCATTER_THREADS) op[j] = s_vals[j];
      }
    }
    if constexpr (HAS_VALID && STAGED) {
      const int64_t g0 = cb & ~63ll;
      const int nw = (lead + cnt + 63) >> 6;  // output words this chunk touches: LDS word pair q IS output word (g0 >> 6) + q
      if (t < nw) {
        const uint64_t word = (uint64_t)s_bits[2 * t] | ((uint64_t)s_bits[2 * t + 1] << 32);
        s_bits[2 * t] = 0;  // (clean for the tile's next chunk)
        s_bits[2 * t + 1] = 0;
        if (word) atomicOr(&c_out_valid[(g0 >> 6) + t], (unsigned long long)word);
        vc += __popcll(word);
      }
      static_assert((CAP + 64 + 63) / 64 <= SCATTER_THREADS && 2 * ((CAP + 64 + 63) / 64) <= NBITS32, "one thread per output word");
    } else if constexpr (HAS_VALID) {
      const int64_t g0 = cb & ~63ll;
      const int span = lead + cnt;
      const int span64 = (span + 63) & ~63;
      for (int q = t; q < span64; q += SCATTER_THREADS) {
        int j = q - lead;
        int f = (j >= 0 && j < cnt) ? (int)s_flag[j] : 0;
        uint64_t word = __ballot(f);
        if (lane == 0) {
          int64_t wi = (g0 + q) >> 6;
          // EVERY word goes through atomicOr (buffer pre-zeroed): tiles share their boundary
          // words, and mixing plain stores with L2 atomics on one cache line costs 0.5 ms
          // per 1e9 rows on MI355X (measured), while atomics alone are free.
          if (word) atomicOr(&c_out_valid[wi], (unsigned long long)word);
          vc += __popcll(word);
        }
      }
    }
    if (p0 + CAP < hi_t) __syncthreads();  // LDS is reused by the next chunk
  }
  if constexpr (HAS_VALID) {
    if (c_valid_slots) {  // (null: a plain result's valid rows are counted from its output bitmap afterwards)
      if constexpr (STAGED) {  // (the staged write-out counts per thread, the ballot form on lane 0 only)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vc += __shfl_xor(vc, o, 64);
      }
      if (lane == 0) s_vc[wave] = (uint32_t)vc;
      __syncthreads();
      if (t == 0) {
        uint32_t c = s_vc[0] + s_vc[1] + s_vc[2] + s_vc[3];
        if (a.nulls_mode) c = (uint32_t)(hi_t - lo_t) - c;  // NULL rows this tile appended
        if (c) atomicAdd(&c_valid_slots[tile & (VALID_SLOTS - 1)], (unsigned long long)c);
      }
    }
  }
}

// XCD-aware tile mapping: workgroup b lands on XCD b % 8 (observed dispatch order, used for speed only), so XCD x
// walks the contiguous tile range [x*per, (x+1)*per): neighbouring tiles, which share output cache lines and boundary
// bitmap words, meet in ONE L2 instead of ping-ponging dirty lines between XCDs.  -1: this workgroup has no tile.
__device__ __forceinline__ int64_t scatter_tile_of_block(int xcd_remap, int64_t ntiles) {
  if (!xcd_remap) return blockIdx.x;
  const int64_t per = (ntiles + 7) >> 3;
  const int64_t tile = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return (tile >= ntiles || (int64_t)(blockIdx.x >> 3) >= per) ? -1 : tile;
}

template <int W, int V, bool HAS_VALID, bool SKIP, int S = 1>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_kernel(ScatterArgs a) {
  // this workgroup's column (uniform: scalar selects)
  const void* c_values = a.values;
  BitView c_vvalid = a.vvalid;
  void* c_out_values = a.out_values;
  unsigned long long* c_out_valid = a.out_valid;
  unsigned long long* c_valid_slots = a.valid_slots;
  if (blockIdx.y) {
    const ScatterArgs::Col& c = a.more[blockIdx.y - 1];
    c_values = c.values;
    c_vvalid = c.vvalid;
    c_out_values = c.out_values;
    c_out_valid = c.out_valid;
    c_valid_slots = c.valid_slots;
  }
  const int64_t tile = scatter_tile_of_block(a.xcd_remap, a.ntiles);
  if (tile < 0) return;
  scatter_tile<W, V, HAS_VALID, SKIP, S>(a, tile, c_values, c_vvalid, c_out_values, c_out_valid, c_valid_slots);
}

// sub-tiles per workgroup of the plain tiled launch, by value width (scatter_tile): 16-32 KiB of values per workgroup
#ifndef AH_SCATTER_S1
#define AH_SCATTER_S1 4
#endif
#ifndef AH_SCATTER_S2
#define AH_SCATTER_S2 2
#endif
#ifndef AH_SCATTER_S4
#define AH_SCATTER_S4 1  // (2 measured: no gain at Int32, 1.607 vs 1.598 ms per 2e9 rows)
#endif
constexpr int scatter_sub_tiles(int w) { return w == 1 ? AH_SCATTER_S1 : w == 2 ? AH_SCATTER_S2 : w == 4 ? AH_SCATTER_S4 : 1; }

// ---- sparse selections (at most two selected rows per predicate word on average: K * 32 <= len).  The tiled kernel above
// is then bound by its own per-tile latency chain, not by bytes: 244 K tiles of a 1e9-row column, each a workgroup that
// loads its words, synchronises, loads a handful of values, synchronises, stores — five workgroups per CU, 0.45-0.75 ms
// at 0.1 % selected where the bytes (the two bitmaps + ~1e6 lines) are worth ~0.1 ms.  Here a tile belongs to ONE WAVE
// (lane l = predicate word l, no LDS, no barrier): the wave scans its 64 popcounts, every lane walks the set bits of its
// own word — up to four value loads in flight — and stores each selected value straight to its output row; validity
// bits are merged with atomicOr (one per valid selected row: few by construction).  32 tiles in flight per CU
// instead of 5.  Same results, window / destination-offset / NULL-counting modes included.
template <int W, bool HAS_VALID>
__device__ __forceinline__ void sparse_tile(const ScatterArgs& a, int64_t tile, int lane, int wave, const void* c_values,
                                            BitView c_vvalid, void* c_out_values, unsigned long long* c_out_valid,
                                            unsigned long long* c_valid_slots) {
  // W == 0: a bit stream alone (Boolean values / a validity bitmap through filter_bits, filter.rs:680-729): the stream
  // travels as `c_vvalid`, there are no value loads or stores
  using ET = typename Elem<W == 0 ? 1 : W>::type;
  static_assert(W != 0 || HAS_VALID, "the bit-only form compacts the HAS_VALID stream");
  constexpr int T = 4096;
  const int64_t row0 = tile * T, s = row0 + ((int64_t)lane << 6);
  uint64_t m = 0, v = 0;
  if (s < a.len) {  // the three bitmaps' words are requested together (one round trip), as in scatter_tile
    const bool has_mv = a.mask_valid.words != nullptr;
    const bool has_vv = HAS_VALID && c_vvalid.words != nullptr;
    const BvRaw rm = bv_issue(a.mask, s, a.len);
    const BvRaw rmv = bv_issue(has_mv ? a.mask_valid : a.mask, s, a.len);
    const BvRaw rvv = bv_issue(has_vv ? c_vvalid : a.mask, s, a.len);
    m = bv_finish(rm, s, a.len);
    const uint64_t mv = bv_finish(rmv, s, a.len), vv = bv_finish(rvv, s, a.len);
    if (has_mv) m &= mv;
    if constexpr (HAS_VALID) v = has_vv ? vv : ~0ull;
  }
  const int c = __popcll(m);
  const int incl = wave_scan_incl(c);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  if (total == 0) return;
  const int64_t chunk0 = a.chunk_base + row0 / CHUNK_ROWS;
  int64_t pos = (int64_t)a.group_prefix[chunk0 >> a.group_shift] + a.chunk_prefix[chunk0] + (incl - c);  // in the filtered stream
  const ET* vp = (const ET*)c_values + s;
  ET* op = (ET*)c_out_values;
  int cnt = 0;  // valid rows written (nulls_mode: NULL rows written)
  // validity: the wave's rows occupy consecutive output positions, i.e. at most 65 output words; their bits are
  // collected in LDS (ds_or, wave-private: no barrier) and leave as ONE global atomicOr per word (one per ROW cost
  // 0.64 ms per 10^6 rows: the output bitmap of a sparse selection is a few hundred cache lines, every atomic lands there)
  __shared__ unsigned long long s_w[HAS_VALID ? SCATTER_THREADS / 64 : 1][HAS_VALID ? 66 : 1];
  const int64_t first = (int64_t)a.group_prefix[chunk0 >> a.group_shift] + a.chunk_prefix[chunk0];  // == pos of lane 0's first row
  int64_t lo = first, hi = first + total;  // this wave's positions, clipped to the launch's window
  if (a.win_hi != 0) {
    lo = lo < a.win_lo ? a.win_lo : lo;
    hi = hi > a.win_hi ? a.win_hi : hi;
  }
  if (lo >= hi) return;
  const int64_t wb = (a.out_base + lo - a.win_lo) >> 6;                       // first output word this wave touches
  const int nw = (int)(((a.out_base + hi - 1 - a.win_lo) >> 6) - wb) + 1;   // <= 65
  if constexpr (HAS_VALID) {
    s_w[wave][lane] = 0;
    if (lane < 2) s_w[wave][64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  while (m) {  // up to four rows per round: their loads go out together
    int b[4], nb = 0;
    ET x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      b[k] = 0;
      if (m) {
        b[k] = __builtin_ctzll(m);
        m &= m - 1;
        nb = k + 1;
        if constexpr (W != 0) x[k] = vp[b[k]];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < nb) {
        const int64_t p = pos + k;
        if (p >= lo && p < hi) {
          const int64_t dst = a.out_base + p - a.win_lo;
          if constexpr (W != 0) op[dst] = x[k];
          if constexpr (HAS_VALID) {
            const int vb = (int)((v >> b[k]) & 1ull);
            if (vb) atomicOr(&s_w[wave][(int)((dst >> 6) - wb)], 1ull << (dst & 63));
            cnt += a.nulls_mode ? 1 - vb : vb;
          }
        }
      }
    }
    pos += nb;
  }
  if constexpr (HAS_VALID) {
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < nw; j += 64) {
      const unsigned long long w = s_w[wave][j];
      if (w) atomicOr(&c_out_valid[wb + j], w);
    }
    if (c_valid_slots) {  // (null: the caller counts the output bitmap afterwards — scatter_valid_count_kernel)
      const int sum = (int)wave_reduce_add64((unsigned long long)cnt);
      if (lane == 0 && sum) atomicAdd(&c_valid_slots[tile & (VALID_SLOTS - 1)], (unsigned long long)sum);
    }
  }
}

template <int W, bool HAS_VALID>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_sparse_kernel(ScatterArgs a) {
  const void* c_values = a.values;
  BitView c_vvalid = a.vvalid;
  void* c_out_values = a.out_values;
  unsigned long long* c_out_valid = a.out_valid;
  unsigned long long* c_valid_slots = a.valid_slots;
  if (blockIdx.y) {
    const ScatterArgs::Col& c = a.more[blockIdx.y - 1];
    c_values = c.values;
    c_vvalid = c.vvalid;
    c_out_values = c.out_values;
    c_out_valid = c.out_valid;
    c_valid_slots = c.valid_slots;
  }
  const int lane = threadIdx.x & 63, wave = ah_uniform((int)(threadIdx.x >> 6));
  const int64_t nwg = (a.ntiles + 3) >> 2;
  const int64_t wg = scatter_tile_of_block(a.xcd_remap, nwg);
  if (wg < 0) return;
  const int64_t tile = wg * 4 + wave;
  if (tile >= a.ntiles) return;
  sparse_tile<W, HAS_VALID>(a, tile, lane, wave, c_values, c_vvalid, c_out_values, c_out_valid, c_valid_slots);
}

// Valid rows of a sparse result, counted from its (small) output bitmap: one atomic per BLOCK into the column's 64
// counters.  A per-tile atomic inside the scatter — 244 K of them on 64 addresses for a 1e9-row column — serialises at
// ~150 ns per same-address atomic across the XCDs: 0.55 ms, which a 1.3 ms dense scatter hides and a 0.1 ms sparse
// one does not.
__global__ void __launch_bounds__(256) scatter_valid_count_kernel(ScatterArgs a, int64_t out_rows) {
  const unsigned long long* bits = a.out_valid;
  unsigned long long* slots = a.valid_slots;
  if (blockIdx.y) {
    bits = a.more[blockIdx.y - 1].out_valid;
    slots = a.more[blockIdx.y - 1].valid_slots;
  }
  const int64_t nwords = (out_rows + 63) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
    unsigned long long x = bits[w];
    if (w == nwords - 1 && (out_rows & 63)) x &= (1ull << (out_rows & 63)) - 1ull;
    acc += (unsigned long long)__popcll(x);
  }
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s_acc[4];
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3];
    if (t) atomicAdd(&slots[blockIdx.x & (VALID_SLOTS - 1)], t);
  }
}

// ---- several input batches, one launch (BatchCoalescer::push_batches_with_filters): the tiles of up to 8 batches form
// one tile space; every batch brings its own predicate tables, source columns and window of its filtered stream, all
// of them append to the SAME in-progress output columns (batch b's rows at out_base[b] onwards).
constexpr int MULTI_MAX_SEGS = 8;
struct MultiSeg {
  BitView mask, mask_valid;
  int64_t len;
  const uint32_t* chunk_prefix;
  const unsigned long long* group_prefix;
  int group_shift;
  int64_t out_base, win_lo, win_hi;
  int64_t tile0;    // first tile of this batch in the launch's tile space
  int64_t tile_lo;  // ... which stands for tile `tile_lo` of the batch (tiles before it hold no row of the window)
  struct Src {
    const void* values;
    BitView vvalid;
  } col[SCATTER_MAX_COLS];
};
struct MultiArgs {
  int nsegs, xcd_remap;
  int64_t ntiles;  // of all segments together
  struct Dst {
    void* out_values;
    unsigned long long* out_valid;
    unsigned long long* null_slots;
  } dst[SCATTER_MAX_COLS];
  MultiSeg seg[MULTI_MAX_SEGS];
};

template <int W, int V, bool SKIP>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_multi_kernel(MultiArgs m) {
  const int64_t gtile = scatter_tile_of_block(m.xcd_remap, m.ntiles);
  if (gtile < 0) return;
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX_SEGS; ++i)
    if (i < m.nsegs && gtile >= m.seg[i].tile0) sidx = i;
  const MultiSeg& sg = m.seg[sidx];
  ScatterArgs a{};
  a.mask = sg.mask;
  a.mask_valid = sg.mask_valid;
  a.len = sg.len;
  a.chunk_prefix = sg.chunk_prefix;
  a.group_prefix = sg.group_prefix;
  a.group_shift = sg.group_shift;
  a.out_base = sg.out_base;
  a.win_lo = sg.win_lo;
  a.win_hi = sg.win_hi;
  a.nulls_mode = 1;
  const MultiSeg::Src& src = sg.col[blockIdx.y];
  const MultiArgs::Dst& d = m.dst[blockIdx.y];
  scatter_tile<W, V, true, SKIP>(a, gtile - sg.tile0 + sg.tile_lo, src.values, src.vvalid, d.out_values, d.out_valid, d.null_slots);
}

// the multi-batch launch for sparse selections: the tile space of all segments, one 4096-row tile per wave (W <= 8: the
// tiled kernels' tile is 4096 rows too, so MultiSeg::tile0 means the same), no counters: the caller counts afterwards
template <int W>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_multi_sparse_kernel(MultiArgs m) {
  static_assert(tile_rows(W) == 4096, "tile0 is counted in 4096-row tiles");
  const int lane = threadIdx.x & 63, wave = ah_uniform((int)(threadIdx.x >> 6));
  const int64_t nwg = (m.ntiles + 3) >> 2;
  const int64_t wg = scatter_tile_of_block(m.xcd_remap, nwg);
  if (wg < 0) return;
  const int64_t gtile = wg * 4 + wave;
  if (gtile >= m.ntiles) return;
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < MULTI_MAX_SEGS; ++i)
    if (i < m.nsegs && gtile >= m.seg[i].tile0) sidx = i;
  const MultiSeg& sg = m.seg[sidx];
  ScatterArgs a{};
  a.mask = sg.mask;
  a.mask_valid = sg.mask_valid;
  a.len = sg.len;
  a.chunk_prefix = sg.chunk_prefix;
  a.group_prefix = sg.group_prefix;
  a.group_shift = sg.group_shift;
  a.out_base = sg.out_base;
  a.win_lo = sg.win_lo;
  a.win_hi = sg.win_hi;
  a.nulls_mode = 1;
  const MultiSeg::Src& src = sg.col[blockIdx.y];
  const MultiArgs::Dst& d = m.dst[blockIdx.y];
  sparse_tile<W, true>(a, gtile - sg.tile0 + sg.tile_lo, lane, wave, src.values, src.vvalid, d.out_values, d.out_valid, nullptr);
}

// ---- batch tables in device memory (filter_internal.hpp: ah_tbl_*).  A count wave owns 64 consecutive chunks of the PUSH —
// across batch boundaries: an 8192-row batch is 8 chunks, and one wave per batch (the first form) left seven eighths of every
// wave idle (0.75 ms per 1e9 rows of 8192-row batches against 0.22 ms at 65 536-row batches).  chunk_seg[c] names chunk c's
// batch; lanes 16g .. 16g + 15 of step `it` read the 16 mask words of chunk 64 w + 4 it + g.  Two halves of eight steps:
// each half's loads are all requested before the first is used.
__global__ void __launch_bounds__(64) filter_count_table_kernel(ah_tbl_push t) {
  __shared__ uint32_t s_cnt[64];
  const int lane = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int64_t c_last = c0 + 63 < t.nchunks ? c0 + 63 : t.nchunks - 1;
  const int seg_first = ah_uniform(t.chunk_seg[c0]), seg_last = ah_uniform(t.chunk_seg[c_last]);
  if (seg_first == seg_last) {
    // the whole wave lies in ONE batch (every wave of a batch of >= 64 chunks but its first and last): the batch's tables are
    // wave-uniform and all 16 steps' words are requested together, as filter_count_small_kernel does — the per-lane table
    // walk below costs three dependent round trips per half (0.35 against 0.22 ms per 1e9 rows at 65 536-row batches)
    const ah_tbl_seg& o = t.segs[seg_first];
    const int64_t chunk_base = c0 - o.chunk0, len = o.len;
    const BitView mask = o.mask, mask_valid = o.mask_valid;
    const bool has_mv = mask_valid.words != nullptr;
    BvRaw rm[16], rv[16] = {};
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
      rm[it] = bv_issue(mask, s < len ? s : 0, len);
    }
    if (has_mv) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
        rv[it] = bv_issue(mask_valid, s < len ? s : 0, len);
      }
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int64_t s = ((chunk_base + it * 4) * 16 + lane) << 6;
      const int64_t sc = s < len ? s : 0;
      uint64_t mk = bv_finish(rm[it], sc, len);
      if (has_mv) mk &= bv_finish(rv[it], sc, len);
      if (s >= len) mk = 0;
      int c = __popcll(mk);
      c += __shfl_xor(c, 1, 64);
      c += __shfl_xor(c, 2, 64);
      c += __shfl_xor(c, 4, 64);
      c += __shfl_xor(c, 8, 64);
      if ((lane & 15) == 0) s_cnt[it * 4 + (lane >> 4)] = (uint32_t)c;
    }
  } else {
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    BvRaw rm[8], rv[8];
    int64_t ss[8], ln[8];
    bool mv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int it = half * 8 + k;
      const int64_t c = c0 + it * 4 + (lane >> 4);
      const bool live = c < t.nchunks;
      const ah_tbl_seg& o = t.segs[live ? t.chunk_seg[c] : 0];
      const int64_t s = live ? (((c - o.chunk0) * 16 + (lane & 15)) << 6) : 0;
      const int64_t len = o.len;
      const bool in = live && s < len;
      mv[k] = o.mask_valid.words != nullptr;
      ss[k] = in ? s : -1;
      ln[k] = len;
      rm[k] = bv_issue(o.mask, in ? s : 0, len);
      rv[k] = bv_issue(mv[k] ? o.mask_valid : o.mask, in ? s : 0, len);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int it = half * 8 + k;
      uint64_t mk = 0;
      if (ss[k] >= 0) {
        mk = bv_finish(rm[k], ss[k], ln[k]);
        const uint64_t v = bv_finish(rv[k], ss[k], ln[k]);
        if (mv[k]) mk &= v;
      }
      int c = __popcll(mk);
      c += __shfl_xor(c, 1, 64);
      c += __shfl_xor(c, 2, 64);
      c += __shfl_xor(c, 4, 64);
      c += __shfl_xor(c, 8, 64);
      if ((lane & 15) == 0) s_cnt[it * 4 + (lane >> 4)] = (uint32_t)c;
    }
  }
  }
  __syncthreads();
  const int v = (int)s_cnt[lane];
  const int incl = wave_scan_incl(v);
  if (c0 + lane < t.nchunks) t.chunk_prefix[c0 + lane] = (uint32_t)(incl - v);
  if (lane == 63) t.wave_total[blockIdx.x] = (uint32_t)incl;
}
// one block: exclusive scan of the wave totals -> wave_prefix[0 .. nwaves] (the last entry is K), in device memory for the
// scatter and in the host's pinned words (what the host waits for; the mailbox is posted by the kernel behind this one)
__global__ void __launch_bounds__(1024) filter_scan_table_kernel(ah_tbl_push t, uint64_t* pin, uint64_t* mail, uint64_t seq) {
  __shared__ unsigned long long s_wave[16];
  __shared__ unsigned long long s_carry;
  const int th = threadIdx.x, lane = th & 63, wave = th >> 6;
  if (th == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < t.nwaves; base += 1024) {
    unsigned long long v = (base + th < t.nwaves) ? t.wave_total[base + th] : 0ull, incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long wbase = s_carry;
    for (int w = 0; w < wave; ++w) wbase += s_wave[w];
    if (base + th < t.nwaves) t.wave_prefix[base + th] = wbase + incl - v;
    __syncthreads();
    if (th == 1023) s_carry = wbase + incl;
    __syncthreads();
  }
  if (th == 0) t.wave_prefix[t.nwaves] = s_carry;
  __threadfence();
  __syncthreads();
  for (int64_t i = th; i <= t.nwaves; i += 1024)
    __hip_atomic_store(pin + i, (uint64_t)t.wave_prefix[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();  // every thread's pinned words before the barrier ...
  __syncthreads();
  if (th == 0) ah_mail_post(mail, seq);  // ... and the sequence word after it (a separate posting kernel cost ~10 us per push)
}

struct TblScatterArgs {
  ah_tbl_push t;
  int64_t tile_lo, ntiles;  // this launch walks tiles [tile_lo, tile_lo + ntiles) of the tile table
  int xcd_remap;
  int64_t win_lo, win_hi, out_base;
  int col_index[AH_TBL_MAX_COLS];
  ah_tbl_dst dst[AH_TBL_MAX_COLS];
};
__device__ __forceinline__ ScatterArgs tbl_tile_args(const TblScatterArgs& m, const ah_tbl_seg& sg) {
  ScatterArgs a{};
  a.mask = sg.mask;
  a.mask_valid = sg.mask_valid;
  a.len = sg.len;
  a.chunk_prefix = m.t.chunk_prefix;  // indexed by the push's global chunk number: chunk_base + the tile's chunk
  a.group_prefix = m.t.wave_prefix;   // positions in the filtered stream of the WHOLE push, one per 64 global chunks
  a.group_shift = 6;
  a.chunk_base = sg.chunk0;
  a.out_base = m.out_base;
  a.win_lo = m.win_lo;
  a.win_hi = m.win_hi;
  return a;
}
template <int W, int V, bool SKIP>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_table_kernel(TblScatterArgs m) {
  const int64_t g = scatter_tile_of_block(m.xcd_remap, m.ntiles);
  if (g < 0) return;
  const ah_tbl_tile tt = m.t.tiles[m.tile_lo + g];
  const ah_tbl_seg& sg = m.t.segs[tt.seg];
  const ScatterArgs a = tbl_tile_args(m, sg);
  const ah_tbl_col sc = m.t.cols[(int64_t)tt.seg * m.t.ncols + m.col_index[blockIdx.y]];
  const ah_tbl_dst d = m.dst[blockIdx.y];
  scatter_tile<W, V, true, SKIP>(a, tt.tile, sc.values, sc.vvalid, d.out_values, d.out_valid, nullptr);
}
template <int W>
__global__ void __launch_bounds__(SCATTER_THREADS) filter_scatter_table_sparse_kernel(TblScatterArgs m) {
  static_assert(tile_rows(W) == 4096, "the tile table is in 4096-row tiles");
  const int lane = threadIdx.x & 63, wave = ah_uniform((int)(threadIdx.x >> 6));
  const int64_t nwg = (m.ntiles + 3) >> 2;
  const int64_t wg = scatter_tile_of_block(m.xcd_remap, nwg);
  if (wg < 0) return;
  const int64_t g = wg * 4 + wave;
  if (g >= m.ntiles) return;
  const ah_tbl_tile tt = m.t.tiles[m.tile_lo + g];
  const ah_tbl_seg& sg = m.t.segs[tt.seg];
  ScatterArgs a = tbl_tile_args(m, sg);
  a.nulls_mode = 1;
  const ah_tbl_col sc = m.t.cols[(int64_t)tt.seg * m.t.ncols + m.col_index[blockIdx.y]];
  const ah_tbl_dst d = m.dst[blockIdx.y];
  sparse_tile<W, true>(a, tt.tile, lane, wave, sc.values, sc.vvalid, d.out_values, d.out_valid, nullptr);
}

// zero bits of `nbatches` consecutive `stride`-bit ranges of up to 8 bitmaps (grid: batch x column; one wave each)
struct NullBatches {
  const unsigned long long* bits[AH_TBL_MAX_COLS];
  int ncols, parts;  // `parts` workgroups share one batch's words
  int64_t stride, nbatches, last_rows;
  unsigned long long* acc;  // device: nbatches x ncols words, zero on entry
};
// (one wave per batch was the first form: a 2^20-row target is 16 K words — 256 dependent iterations per wave, 7 ms per 1e9 rows)
__global__ void __launch_bounds__(256) count_nulls_batches_kernel(NullBatches nb) {
  const int64_t j = blockIdx.x / nb.parts;
  const int part = (int)(blockIdx.x % nb.parts);
  const unsigned long long* bits = nb.bits[blockIdx.y] + j * (nb.stride >> 6);
  const int64_t rows = j == nb.nbatches - 1 ? nb.last_rows : nb.stride, nwords = (rows + 63) >> 6;
  unsigned long long nulls = 0;
  for (int64_t w = (int64_t)part * 256 + threadIdx.x; w < nwords; w += (int64_t)nb.parts * 256) {
    unsigned long long in = ~0ull;
    if (w == nwords - 1 && (rows & 63)) in = (1ull << (rows & 63)) - 1ull;
    nulls += (unsigned long long)__popcll(in & ~bits[w]);
  }
  nulls = wave_reduce_add64(nulls);
  if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(nb.acc + j * nb.ncols + blockIdx.y, nulls);
}
// one block: the counted words to the host's pinned words, then the mailbox post
__global__ void __launch_bounds__(1024) words_to_pinned_kernel(const unsigned long long* src, int64_t n, unsigned long long* dst, uint64_t* mail,
                                                               uint64_t seq) {
  for (int64_t i = threadIdx.x; i < n; i += 1024) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) ah_mail_post(mail, seq);
}

// NULL rows among destination rows [bit_lo, bit_lo + nbits) of up to 8 columns (blockIdx.y), added to each column's
// counters: what the sparse multi-batch launch appended, counted from the in-progress bitmaps afterwards
struct RangeCount {
  const unsigned long long* bits[SCATTER_MAX_COLS];
  unsigned long long* slots[SCATTER_MAX_COLS];
  int64_t bit_lo, nbits;
};
__global__ void __launch_bounds__(256) range_null_count_kernel(RangeCount r) {
  const unsigned long long* bits = r.bits[blockIdx.y];
  const int64_t w0 = r.bit_lo >> 6, w1 = (r.bit_lo + r.nbits - 1) >> 6;
  unsigned long long nulls = 0;
  for (int64_t w = w0 + (int64_t)blockIdx.x * 256 + threadIdx.x; w <= w1; w += (int64_t)gridDim.x * 256) {
    unsigned long long in = ~0ull;  // the bits of word w inside the range
    if (w == w0) in &= ~0ull << (r.bit_lo & 63);
    if (w == w1 && ((r.bit_lo + r.nbits) & 63)) in &= (1ull << ((r.bit_lo + r.nbits) & 63)) - 1ull;
    nulls += (unsigned long long)__popcll(in & ~bits[w]);
  }
  nulls = wave_reduce_add64(nulls);
  __shared__ unsigned long long s_acc[4];
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = nulls;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3];
    if (t) atomicAdd(&r.slots[blockIdx.y][blockIdx.x & (VALID_SLOTS - 1)], t);
  }
}

template <int W, bool HV>
void launch_scatter_w(ah_context* ctx, const ScatterArgs& a_in, bool aligned16, bool skip, int ncols = 1, bool sparse = false,
                      int64_t out_rows = 0) {
  constexpr int WE = W == 0 ? 1 : W;
  constexpr int VV = (W == 0) ? 16 : (W >= 16 ? 1 : 16 / W);
  constexpr int SS = W == 0 ? 1 : scatter_sub_tiles(W);
  const int S = (aligned16 || VV == 1) ? SS : 1;  // (the element-wise form for unaligned buffers keeps the 4096-row tile)
  const int T = tile_rows(WE) * S;
  int64_t ntiles = ah_ceil_div(a_in.len, T);
  ScatterArgs a = a_in;
  a.ntiles = ntiles;
  static const char* xr = getenv("AH_FILTER_XCD");
  a.xcd_remap = (xr && xr[0] == '0') ? 0 : 1;
  // Plain results (whole filtered stream, destination row 0, VALID-row counting, K known): the valid rows are counted
  // from the output bitmap AFTER the scatter (scatter_valid_count_kernel: K / 8 bytes, one atomic per block) instead of one
  // atomic per tile on the column's 64 counters — 244 K same-address atomics per 1e9 rows serialise across the XCDs at
  // ~150 ns each (0.55 ms), which shows as soon as the scatter itself is faster than that (3 % selected: 1.02 -> 0.72 ms).
  // (up to ~12 % selected; a denser scatter runs long enough to hide its counter atomics, and counting a 125 MB bitmap
  // afterwards would only add to it)
  const bool plain = !a.nulls_mode && a.out_base == 0 && a.win_hi == 0;
  const bool no_count = HV && plain && a_in.valid_slots == nullptr && ncols == 1;  // (deferred results: nobody wants the count)
  const bool count_after = HV && !no_count && out_rows > 0 && out_rows * 8 <= a_in.len && plain;
  auto launch_count = [&]() {
    const int64_t nwords = (out_rows + 63) >> 6;
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ah_ceil_div(nwords, 256 * 4)));
    scatter_valid_count_kernel<<<dim3(gx, (unsigned)ncols), 256, 0, ctx->stream>>>(a, out_rows);
  };
  ScatterArgs b = a;  // the scatter's copy: no counters when they are filled afterwards
  if (count_after) {
    b.valid_slots = nullptr;
    for (int c = 1; c < ncols; ++c) b.more[c - 1].valid_slots = nullptr;
  }
  if constexpr (W != 0 || HV) {
    // the sparse form serves plain results; the windowed / NULL-counting launches of the coalescer keep the tiled kernel
    // (W == 0, round 4: the bit-only stream of filter_boolean / filter_nulls takes it too)
    if (sparse && (!HV || count_after || no_count)) {
      b.ntiles = ah_ceil_div(a_in.len, 4096);  // one 4096-row tile per wave, four per workgroup
      const int64_t nwg = (b.ntiles + 3) >> 2;
      dim3 g((unsigned)(a.xcd_remap ? 8 * ((nwg + 7) / 8) : nwg), (unsigned)ncols);
      filter_scatter_sparse_kernel<W, HV><<<g, SCATTER_THREADS, 0, ctx->stream>>>(b);
      if (count_after) launch_count();
      return;
    }
  }
  dim3 grid((unsigned)(a.xcd_remap ? 8 * ((ntiles + 7) / 8) : ntiles), (unsigned)ncols), block(SCATTER_THREADS);
  if constexpr (W == 0) {
    filter_scatter_kernel<W, VV, HV, false><<<grid, block, 0, ctx->stream>>>(b);
  } else if (aligned16 || VV == 1) {
    // (1- and 2-byte values: a 128-byte line holds 128 / 64 rows, so no line goes unselected before the sparse kernel
    // takes over at ~3 % — predicating the loads would only serialise them)
    if constexpr (W <= 2) filter_scatter_kernel<W, VV, HV, false, SS><<<grid, block, 0, ctx->stream>>>(b);
    else if (skip) filter_scatter_kernel<W, VV, HV, true, SS><<<grid, block, 0, ctx->stream>>>(b);
    else filter_scatter_kernel<W, VV, HV, false, SS><<<grid, block, 0, ctx->stream>>>(b);
  } else {
    filter_scatter_kernel<W, 1, HV, false><<<grid, block, 0, ctx->stream>>>(b);
  }
  if (count_after) launch_count();
}

template <bool HV>
ah_status launch_scatter(ah_context* ctx, int width, const ScatterArgs& a, bool skip, int ncols = 1, bool sparse = false,
                         int64_t out_rows = 0) {
  bool aligned16 = (((uintptr_t)a.values) & 15) == 0;
  for (int c = 1; c < ncols; ++c) aligned16 = aligned16 && (((uintptr_t)a.more[c - 1].values) & 15) == 0;
  switch (width) {
    case 0: launch_scatter_w<0, HV>(ctx, a, true, false, ncols, sparse, out_rows); break;
    case 1: launch_scatter_w<1, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    case 2: launch_scatter_w<2, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    case 4: launch_scatter_w<4, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    case 8: launch_scatter_w<8, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    case 16: launch_scatter_w<16, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    case 32: launch_scatter_w<32, HV>(ctx, a, aligned16, skip, ncols, sparse, out_rows); break;
    default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
  }
  return AH_OK;
}

// at most two selected rows per predicate word on average: the wave-per-tile kernel (AH_FILTER_SPARSE=0 / 1 force)
bool use_sparse(int64_t count, int64_t len) {
  const char* env = getenv("AH_FILTER_SPARSE");  // read per call: tests and the A/B flip it
  if (env && env[0] == '0') return false;
  if (env && env[0] == '1') return true;
  return count * 32 <= len;  // crossover measured between 3 % and 10 % selected (profiles/r03_selectivity_sweep.md)
}

// the bit-only stream (filter_boolean, filter_bits): no value traffic, so the wave-per-tile form wins further up —
// measured on 1e9 Boolean rows with 10 % nulls (profiles/r04_bool_filter_ab.md): 0.43 against 0.85 ms at 10 % selected
bool use_sparse_bits(int64_t count, int64_t len) {
  const char* env = getenv("AH_FILTER_SPARSE");
  if (env && env[0] == '0') return false;
  if (env && env[0] == '1') return true;
  return count * 8 <= len;
}

// load predication pays when most 128-byte lines hold no selected row; tunable for experiments
bool use_skip(int64_t count, int64_t len) {
  static const char* env = getenv("AH_FILTER_SKIP");  // "0" / "1" force, unset = heuristic
  if (env && env[0] == '0') return false;
  if (env && env[0] == '1') return true;
  return (double)count < 0.12 * (double)len;
}

}  // namespace

// (struct ah_filter_predicate: filter_internal.hpp)

// K2 for filter_expr.hip's own count pass
void ah_filter_launch_group_scan(ah_context* ctx, const uint32_t* group_total, int64_t ngroups, unsigned long long* group_prefix,
                                 unsigned long long* total, int slot, uint64_t seq) {
  filter_group_scan_kernel<<<1, 1024, 0, ctx->stream>>>(group_total, ngroups, group_prefix, total, ctx->pinned_dev, slot, seq);
}

// The count pass of a predicate, enqueued only: K lands in pinned slot `slot` (and, with seq != 0, the mailbox is
// posted).  `*enqueued` = false when there was nothing to launch (empty predicate: count 0).
static ah_status predicate_enqueue(ah_context* ctx, const ah_array_view* predicate, int slot, uint64_t seq,
                                   ah_filter_predicate** out, bool* enqueued) {
  *out = nullptr;
  *enqueued = false;
  if (predicate->type != AH_BOOL)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "filter predicate must be Boolean, got %s",
                   ah_type_name(predicate->type));
  hipSetDevice(ctx->device);
  auto* p = new ah_filter_predicate();
  p->len = predicate->length;
  p->mask = make_bitview(predicate->values, predicate->values_bit_offset);
  // FilterBuilder::new_with_count (filter.rs:260-273): nulls -> false
  if (predicate->validity && predicate->null_count != 0)
    p->mask_valid = make_bitview(predicate->validity, predicate->validity_bit_offset);
  else
    p->mask_valid = BitView{nullptr, 0};
  if (p->len <= 0) {
    p->len = 0;
    *out = p;
    return AH_OK;
  }
  *enqueued = true;
  int64_t nchunks = ah_ceil_div(p->len, CHUNK_ROWS);
  const bool small = nchunks <= 65536;  // <= 64 Mi rows: one-wave count blocks, 64-chunk groups
  p->group_shift = small ? 6 : 10;
  int64_t ngroups = ah_ceil_div(nchunks, small ? 64 : GROUP_CHUNKS);
  size_t b_chunk = ((size_t)nchunks * 4 + 255) & ~(size_t)255;
  size_t b_gt = ((size_t)ngroups * 4 + 255) & ~(size_t)255;
  size_t b_gp = ((size_t)ngroups * 8 + 255) & ~(size_t)255;
  ah_status st = ah_pool_alloc(ctx, b_chunk + b_gt + b_gp + 256, &p->block);
  if (st != AH_OK) {
    delete p;
    return st;
  }
  char* base = (char*)p->block;
  p->chunk_prefix = (uint32_t*)base;
  uint32_t* group_total = (uint32_t*)(base + b_chunk);
  p->group_prefix = (unsigned long long*)(base + b_chunk + b_gt);
  unsigned long long* total = (unsigned long long*)(base + b_chunk + b_gt + b_gp);
  p->total_dev = total;
  {
    ah_prof_scope ps(ctx, "filter_count");
    if (small) {
      filter_count_small_kernel<<<(unsigned)ngroups, 64, 0, ctx->stream>>>(p->mask, p->mask_valid, p->len, p->chunk_prefix,
                                                                          group_total);
    } else {
      filter_count_kernel<<<(unsigned)ngroups, 1024, 0, ctx->stream>>>(p->mask, p->mask_valid, p->len,
                                                                      p->chunk_prefix, group_total);
    }
    filter_group_scan_kernel<<<1, 1024, 0, ctx->stream>>>(group_total, ngroups, p->group_prefix,
                                                          total, ctx->pinned_dev, slot, seq);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ah_pool_free(ctx, p->block);
    delete p;
    return ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  }
  *out = p;
  return AH_OK;
}

extern "C" ah_status ah_filter_predicate_build(ah_context* ctx, const ah_array_view* predicate,
                                               ah_filter_predicate** out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !predicate || !out) return AH_INVALID_ARGUMENT;
  const uint64_t seq = ah_mail_next(ctx);
  bool enqueued = false;
  ah_filter_predicate* p = nullptr;
  *out = nullptr;
  AH_TRY(predicate_enqueue(ctx, predicate, 0, seq, &p, &enqueued));
  if (enqueued) {
    hipError_t e = ah_mail_wait(ctx, seq);
    if (e != hipSuccess) {
      ah_pool_free(ctx, p->block);
      delete p;
      return ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
    }
    p->count = (int64_t)ctx->pinned[0];
  }
  *out = p;
  return AH_OK;
}

// The two halves of ah_filter_predicate_build for a caller that has device work to enqueue between them
// (BatchCoalescer's speculative scatter): begin enqueues the count pass, end waits for K.
ah_status ah_filter_predicate_begin(ah_context* ctx, const ah_array_view* predicate, ah_filter_predicate** out, uint64_t* seq,
                                    bool* enqueued) {
  *seq = ah_mail_next(ctx);
  return predicate_enqueue(ctx, predicate, 0, *seq, out, enqueued);
}
ah_status ah_filter_predicate_end(ah_context* ctx, ah_filter_predicate* p, uint64_t seq, bool enqueued) {
  if (!enqueued) return AH_OK;
  hipError_t e = ah_mail_wait(ctx, seq);
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  p->count = (int64_t)ctx->pinned[0];
  return AH_OK;
}

// Counts of `n` (<= 128) predicates with ONE host wait: every count pass is enqueued, the last mailbox kernel posts for
// all of them (BatchCoalescer's multi-batch push; a host that has several batches queued hides the count round trip).
extern "C" ah_status ah_filter_predicates_build(ah_context* ctx, int32_t n, const ah_array_view* predicates,
                                                ah_filter_predicate** outs) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || n < 0 || n > 128 || (n > 0 && (!predicates || !outs))) return AH_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i) outs[i] = nullptr;
  std::vector<char> launched((size_t)std::max(n, 1), 0);
  ah_status st = AH_OK;
  // small predicates (<= 1 Mi rows... up to 64 Mi: <= 1024 count groups) are counted eight per launch
  CountMulti cm{};
  int ncm = 0, cm_first = 0;
  int64_t cm_groups = 0;
  auto flush_multi = [&]() {
    if (ncm == 0) return;
    ah_prof_scope ps(ctx, "filter_count");
    filter_count_small_multi_kernel<<<dim3((unsigned)cm_groups, (unsigned)ncm), 64, 0, ctx->stream>>>(cm);
    filter_group_scan_multi_kernel<<<(unsigned)ncm, 1024, 0, ctx->stream>>>(cm, ctx->pinned_dev, 16 + cm_first);
    ncm = 0;
    cm_groups = 0;
  };
  for (int i = 0; i < n && st == AH_OK; ++i) {
    const ah_array_view* pv = &predicates[i];
    const int64_t nchunks = pv->type == AH_BOOL && pv->length > 0 ? ah_ceil_div(pv->length, CHUNK_ROWS) : 0;
    if (nchunks == 0 || nchunks > 65536) {  // empty / wrong type (error below) / large: the single-predicate path
      flush_multi();
      bool enq = false;
      st = predicate_enqueue(ctx, pv, 16 + i, 0, &outs[i], &enq);
      launched[i] = enq ? 1 : 0;
      continue;
    }
    auto* p = new ah_filter_predicate();
    p->len = pv->length;
    p->mask = make_bitview(pv->values, pv->values_bit_offset);
    p->mask_valid = (pv->validity && pv->null_count != 0) ? make_bitview(pv->validity, pv->validity_bit_offset) : BitView{nullptr, 0};
    p->group_shift = 6;
    const int64_t ngroups = ah_ceil_div(nchunks, 64);
    const size_t b_chunk = ((size_t)nchunks * 4 + 255) & ~(size_t)255, b_gt = ((size_t)ngroups * 4 + 255) & ~(size_t)255,
                 b_gp = ((size_t)ngroups * 8 + 255) & ~(size_t)255;
    st = ah_pool_alloc(ctx, b_chunk + b_gt + b_gp + 256, &p->block);
    if (st != AH_OK) {
      delete p;
      break;
    }
    char* base = (char*)p->block;
    p->chunk_prefix = (uint32_t*)base;
    p->group_prefix = (unsigned long long*)(base + b_chunk + b_gt);
    p->total_dev = (unsigned long long*)(base + b_chunk + b_gt + b_gp);
    outs[i] = p;
    launched[i] = 1;
    if (ncm == 0) cm_first = i;
    CountMulti::One& o = cm.p[ncm++];
    o.mask = p->mask, o.mask_valid = p->mask_valid, o.len = p->len, o.chunk_prefix = p->chunk_prefix;
    o.group_total = (uint32_t*)(base + b_chunk), o.group_prefix = p->group_prefix, o.total = p->total_dev, o.ngroups = ngroups;
    cm_groups = std::max(cm_groups, ngroups);
    if (ncm == 8) flush_multi();
  }
  if (st == AH_OK) flush_multi();
  if (st == AH_OK) {
    hipError_t e = ah_stream_wait(ctx);  // one flag kernel behind all the count passes
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  }
  if (st != AH_OK) {
    for (int i = 0; i < n; ++i) {
      ah_filter_predicate_free(ctx, outs[i]);
      outs[i] = nullptr;
    }
    return st;
  }
  for (int i = 0; i < n; ++i)
    if (launched[i]) outs[i]->count = (int64_t)ctx->pinned[16 + i];
  return AH_OK;
}

// ah_filter_predicates_build in two halves, for a caller that wants its OWN count slots so that two groups can be in flight
// (BatchCoalescer's begin / end pushes): `begin` enqueues the count passes of n (<= 64) predicates whose K's land in
// pin_dev[0 .. n) (the device view of the caller's pinned words) and posts the mailbox behind them (*seq); `end` waits for
// that sequence number and fills in the counts from the host view.  Only the multi-count shape (non-empty Boolean
// predicates of at most 64 Mi rows): anything else -> AH_NOT_YET_IMPLEMENTED with nothing enqueued (use the one-call build).
// quant_dev / quant_host (optional): n x AH_FILTER_QUANTS pinned words for the predicates' quantile prefixes
// (ah_filter_predicate::quant).
ah_status ah_filter_predicates_begin(ah_context* ctx, int32_t n, const ah_array_view* predicates, ah_filter_predicate** outs,
                                     uint64_t* pin_dev, uint64_t* seq, uint64_t* quant_dev) {
  if (n < 1 || n > 64) return AH_NOT_YET_IMPLEMENTED;
  for (int i = 0; i < n; ++i) {
    outs[i] = nullptr;
    const ah_array_view* pv = &predicates[i];
    const int64_t nchunks = pv->type == AH_BOOL && pv->length > 0 ? ah_ceil_div(pv->length, CHUNK_ROWS) : 0;
    if (nchunks == 0 || nchunks > 65536) return AH_NOT_YET_IMPLEMENTED;
  }
  ah_status st = AH_OK;
  CountMulti cm{};
  int ncm = 0, cm_first = 0;
  int64_t cm_groups = 0;
  auto flush_multi = [&]() {
    if (ncm == 0) return;
    ah_prof_scope ps(ctx, "filter_count");
    filter_count_small_multi_kernel<<<dim3((unsigned)cm_groups, (unsigned)ncm), 64, 0, ctx->stream>>>(cm);
    filter_group_scan_multi_kernel<<<(unsigned)ncm, 1024, 0, ctx->stream>>>(cm, pin_dev, cm_first);
    ncm = 0;
    cm_groups = 0;
  };
  for (int i = 0; i < n && st == AH_OK; ++i) {
    const ah_array_view* pv = &predicates[i];
    const int64_t nchunks = ah_ceil_div(pv->length, CHUNK_ROWS);
    auto* p = new ah_filter_predicate();
    p->len = pv->length;
    p->mask = make_bitview(pv->values, pv->values_bit_offset);
    p->mask_valid = (pv->validity && pv->null_count != 0) ? make_bitview(pv->validity, pv->validity_bit_offset) : BitView{nullptr, 0};
    p->group_shift = 6;
    const int64_t ngroups = ah_ceil_div(nchunks, 64);
    const size_t b_chunk = ((size_t)nchunks * 4 + 255) & ~(size_t)255, b_gt = ((size_t)ngroups * 4 + 255) & ~(size_t)255,
                 b_gp = ((size_t)ngroups * 8 + 255) & ~(size_t)255;
    st = ah_pool_alloc(ctx, b_chunk + b_gt + b_gp + 256, &p->block);
    if (st != AH_OK) {
      delete p;
      break;
    }
    char* base = (char*)p->block;
    p->chunk_prefix = (uint32_t*)base;
    p->group_prefix = (unsigned long long*)(base + b_chunk + b_gt);
    p->total_dev = (unsigned long long*)(base + b_chunk + b_gt + b_gp);
    outs[i] = p;
    if (ncm == 0) cm_first = i;
    CountMulti::One& o = cm.p[ncm++];
    o.mask = p->mask, o.mask_valid = p->mask_valid, o.len = p->len, o.chunk_prefix = p->chunk_prefix;
    o.group_total = (uint32_t*)(base + b_chunk), o.group_prefix = p->group_prefix, o.total = p->total_dev, o.ngroups = ngroups;
    if (quant_dev) {
      p->quant_step = ah_ceil_div(ngroups, (int64_t)AH_FILTER_QUANTS);
      p->quant_n = (int)ah_ceil_div(ngroups, p->quant_step);
      o.quant_pin = quant_dev + (size_t)i * AH_FILTER_QUANTS, o.quant_step = p->quant_step;
    }
    cm_groups = std::max(cm_groups, ngroups);
    if (ncm == 8) flush_multi();
  }
  if (st == AH_OK) {
    flush_multi();
    ctx->inflight = true;
    hipError_t e = ah_mail_post_async(ctx, seq);
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  }
  if (st != AH_OK) {
    (void)ah_stream_wait(ctx);  // count kernels of earlier predicates may be running on the blocks freed below
    for (int i = 0; i < n; ++i) {
      ah_filter_predicate_free(ctx, outs[i]);
      outs[i] = nullptr;
    }
  }
  return st;
}
ah_status ah_filter_predicates_end(ah_context* ctx, int32_t n, ah_filter_predicate** outs, const uint64_t* pin_host, uint64_t seq,
                                   const uint64_t* quant_host) {
  const bool later_work = seq != ctx->mail_seq || ctx->inflight;  // enqueued behind the post: still running afterwards
  hipError_t e = ah_mail_wait(ctx, seq);
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  if (later_work && ctx->wait_mode != 1) ctx->inflight = true;  // ah_mail_wait cleared it for the work BEFORE the post only
  for (int i = 0; i < n; ++i) {
    ah_filter_predicate* p = outs[i];
    p->count = (int64_t)__atomic_load_n(&pin_host[i], __ATOMIC_RELAXED);
    if (!quant_host) p->quant_n = 0;
    for (int k = 0; k < p->quant_n; ++k) p->quant[k] = __atomic_load_n(&quant_host[(size_t)i * AH_FILTER_QUANTS + k], __ATOMIC_RELAXED);
    p->quant[p->quant_n] = (uint64_t)p->count;
  }
  return AH_OK;
}

extern "C" int64_t ah_filter_predicate_count(const ah_filter_predicate* p) { return p ? p->count : 0; }

extern "C" void ah_filter_predicate_free(ah_context* ctx, ah_filter_predicate* p) {
  ah_ctx_guard _guard(ctx);
  if (!p) return;
  ah_pool_free(ctx, p->block);
  delete p;
}

// compaction of one bit stream (Boolean values or a validity bitmap) by the
// predicate: filter_bits (filter.rs:680-720).  Returns popcount of the result.
static ah_status compact_bits(ah_context* ctx, const ah_filter_predicate* p, BitView src,
                              uint8_t** out_bits, size_t* out_bytes, int64_t* set_bits, bool defer = false) {
  size_t bytes = ah_bitmap_bytes(p->count);
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, &ob));
  unsigned long long* slots = ctx->scratch;  // zero between calls (filter_finish_kernel restores it)
  hipMemsetAsync(ob, 0, bytes, ctx->stream);
  ScatterArgs a{};
  a.values = nullptr;
  a.mask = p->mask;
  a.mask_valid = p->mask_valid;
  a.vvalid = src;
  a.len = p->len;
  a.chunk_prefix = p->chunk_prefix;
  a.group_prefix = p->group_prefix;
  a.group_shift = p->group_shift;
  a.out_values = nullptr;
  a.out_valid = (unsigned long long*)ob;
  a.valid_slots = defer ? nullptr : slots;  // (deferred: the count is reported as unknown, nothing to fold or restore)
  // a sparse selection (<= 1 selected row in 32): the wave-per-tile kernel, set bits counted from the small output
  // bitmap afterwards — as for fixed-width values; filter_boolean was the one value kind left on the tiled
  // kernel at low selectivity
  launch_scatter<true>(ctx, 0, a, false, 1, use_sparse_bits(p->count, p->len), p->count);
  if (defer) {  // no read-back: the caller reports the count as unknown
    *set_bits = -1;
    *out_bits = (uint8_t*)ob;
    *out_bytes = bytes;
    return AH_OK;
  }
  const uint64_t seq = ah_mail_next(ctx);
  filter_finish_kernel<<<1, 64, 0, ctx->stream>>>(slots, ctx->pinned_dev, seq);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  if (e != hipSuccess) {
    ah_out_free(ctx, ob, bytes);
    return ah_fail(ctx, AH_HIP_ERROR, "filter_bits failed: %s", hipGetErrorString(e));
  }
  *set_bits = (int64_t)ctx->pinned[0];
  *out_bits = (uint8_t*)ob;
  *out_bytes = bytes;
  return AH_OK;
}

extern "C" ah_status ah_filter_predicate_apply(ah_context* ctx, const ah_filter_predicate* p,
                                               const ah_array_view* values, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !p || !values || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  // filter_array (filter.rs:535-541)
  if (p->len > values->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT,
                   "Filter predicate of length %lld is larger than target array of length %lld",
                   (long long)p->len, (long long)values->length);
  const bool is_string = values->type == AH_UTF8 || values->type == AH_LARGE_UTF8;
  const int width = is_string ? 0 : ah_type_width(values->type);
  if (width < 0)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "filter not supported for type %s",
                   ah_type_name(values->type));
  out->type = values->type;
  const int64_t K = p->count;
  // deferred mode covers fixed-width and Boolean values: K is known from the predicate, so nothing
  // needs reading back; strings size their byte buffer from the data and stay synchronous
  const bool defer = ctx->deferred && !is_string;
  if (is_string && !values->offsets)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
  if (is_string && (p->len == 0 || K == 0)) {  // new_empty_array: offsets = [0]
    const size_t ow = values->type == AH_UTF8 ? 4 : 8;
    void* offs = nullptr;
    AH_TRY(ah_out_alloc(ctx, ow, &offs));
    hipMemsetAsync(offs, 0, ow, ctx->stream);
    AH_HIP(ctx, ah_stream_wait(ctx));
    out->offsets = offs;
    out->offsets_bytes = (int64_t)ow;
    return AH_OK;
  }
  // IterationStrategy::default_strategy (filter.rs:346-364)
  if (p->len == 0 || K == 0) {  // None -> new_empty_array(data_type) :545
    out->length = 0;
    return AH_OK;
  }
  if (K == p->len) {  // All -> values.slice(0, count) :546 (zero-copy)
    out->length = K;
    out->values = const_cast<void*>(values->values);
    out->values_bit_offset = values->values_bit_offset;
    out->values_bytes = width ? K * width : 0;
    out->offsets = const_cast<void*>(values->offsets);  // strings: the same offsets, first K+1 entries
    out->flags = AH_OUT_BORROWED;
    if (values->validity) {
      int64_t nulls = 0;
      if (K == values->length && values->null_count >= 0) {
        nulls = values->null_count;
      } else if (defer) {
        nulls = -1;
      } else {
        int64_t set = 0;
        AH_TRY(ah_count_set_bits(ctx, values->validity, values->validity_bit_offset, K, &set));
        nulls = K - set;
      }
      out->validity = const_cast<uint8_t*>(values->validity);
      out->validity_bit_offset = values->validity_bit_offset;
      out->null_count = nulls;
    }
    return AH_OK;
  }

  int64_t in_nulls = 0;
  if (defer && values->null_count < 0) in_nulls = values->validity ? 1 : 0;  // unknown: treat as nullable, no count
  else AH_TRY(ah_resolve_null_count(ctx, values, &in_nulls));
  const bool has_valid = values->validity && in_nulls > 0;  // filter_nulls :512-517
  BitView vvalid = has_valid ? make_bitview(values->validity, values->validity_bit_offset)
                             : BitView{nullptr, 0};

  if (is_string) {  // filter_bytes (filter.rs:890-928) + filter_nulls: strings.hip — ranges, tile scan, gather
    const ah_status st = ah_string_filter_bytes(ctx, p, values, vvalid, out);
    if (st != AH_OK) {
      ah_out_init(out);
      return st;
    }
    out->type = values->type;
    out->length = K;
    return AH_OK;
  }

  if (values->type == AH_BOOL) {  // filter_boolean (filter.rs:723-729)
    uint8_t* vb = nullptr;
    size_t vbytes = 0;
    int64_t set = 0;
    AH_TRY(compact_bits(ctx, p, make_bitview(values->values, values->values_bit_offset), &vb,
                        &vbytes, &set, defer));
    out->values = vb;
    out->values_bytes = (int64_t)vbytes;
    out->length = K;
    if (has_valid) {
      uint8_t* nb = nullptr;
      size_t nbytes = 0;
      int64_t nset = 0;
      ah_status st = compact_bits(ctx, p, vvalid, &nb, &nbytes, &nset, defer);
      if (st != AH_OK) {
        ah_array_release(ctx, out);
        return st;
      }
      if (!defer && K - nset == 0) {
        ah_out_free(ctx, nb, nbytes);
      } else {
        out->validity = nb;
        out->validity_bytes = (int64_t)nbytes;
        out->null_count = defer ? -1 : K - nset;
      }
    }
    return AH_OK;
  }

  // filter_primitive (filter.rs:773-788)
  void* ov = nullptr;
  size_t vbytes = (size_t)K * width;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  void* ob = nullptr;
  size_t bbytes = 0;
  unsigned long long* slots = nullptr;
  if (has_valid) {
    bbytes = ah_bitmap_bytes(K);
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st != AH_OK) {
      ah_out_free(ctx, ov, vbytes);
      return st;
    }
    // deferred mode reports null_count = -1 ("unknown"): nothing counts the valid rows, so there are no counters to fold
    // and restore afterwards — one launch less per call (host launch cost is what bounds a batch of small deferred calls)
    slots = defer ? nullptr : ctx->scratch;  // zero between calls (filter_finish_kernel restores it)
    hipMemsetAsync(ob, 0, bbytes, ctx->stream);
  }
  ScatterArgs a{};
  a.values = values->values;
  a.mask = p->mask;
  a.mask_valid = p->mask_valid;
  a.vvalid = vvalid;
  a.len = p->len;
  a.chunk_prefix = p->chunk_prefix;
  a.group_prefix = p->group_prefix;
  a.group_shift = p->group_shift;
  a.out_values = ov;
  a.out_valid = (unsigned long long*)ob;
  a.valid_slots = slots;
  const bool skip = use_skip(K, p->len);
  {
    ah_prof_scope ps(ctx, "filter_scatter");
    if (has_valid) launch_scatter<true>(ctx, width, a, skip, 1, use_sparse(K, p->len), K);
    else launch_scatter<false>(ctx, width, a, skip, 1, use_sparse(K, p->len), K);
  }
  // ONE host wait for the whole scatter: the finish kernel folds the valid-row counters, restores them to
  // zero and posts the mailbox (no D2H copy engine, no hipStreamSynchronize)
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && has_valid && !defer) {
    const uint64_t seq = ah_mail_next(ctx);
    filter_finish_kernel<<<1, 64, 0, ctx->stream>>>(slots, ctx->pinned_dev, seq);
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  } else if (e == hipSuccess && !defer) {
    e = ah_stream_wait(ctx);
  }
  if (e != hipSuccess) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    return ah_fail(ctx, AH_HIP_ERROR, "filter scatter failed: %s", hipGetErrorString(e));
  }
  out->length = K;
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (has_valid && defer) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = -1;
  } else if (has_valid) {
    int64_t nulls = K - (int64_t)ctx->pinned[0];
    if (nulls == 0) {  // filter_nulls :523-525 -> None
      ah_out_free(ctx, ob, bbytes);
    } else {
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = nulls;
    }
  }
  return AH_OK;
}

// InProgressPrimitiveArray::copy_rows_by_filter_from (arrow-select/src/coalesce/primitive.rs): the
// filter scatters straight into an in-progress builder at row `dst_row_offset` — no intermediate
// array (BatchCoalescer::push_batch_with_filter, arrow-select/src/coalesce.rs:229).  The bitmap
// words are merged with atomicOr, so any destination bit offset works.
static ah_status apply_into_impl(ah_context* ctx, const ah_filter_predicate* p, const ah_array_view* values,
                                 void* dst_values, uint8_t* dst_validity, int64_t dst_row_offset,
                                 int64_t* appended_nulls, unsigned long long* nulls_acc, bool no_wait) {
  if (!ctx || !p || !values || !dst_values || !dst_validity) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  if (appended_nulls) *appended_nulls = 0;
  if (p->len > values->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT,
                   "Filter predicate of length %lld is larger than target array of length %lld",
                   (long long)p->len, (long long)values->length);
  const int width = ah_type_width(values->type);
  if (width <= 0)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "fused filter copy not supported for type %s",
                   ah_type_name(values->type));
  if (((uintptr_t)dst_validity & 7) != 0)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "builder validity must be 8-byte aligned");
  const int64_t K = p->count;
  if (p->len == 0 || K == 0) return AH_OK;
  int64_t in_nulls = 0;
  if (no_wait && values->null_count < 0) in_nulls = values->validity ? 1 : 0;  // unknown: treat as nullable, no count
  else AH_TRY(ah_resolve_null_count(ctx, values, &in_nulls));
  const bool has_valid = values->validity && in_nulls > 0;
  unsigned long long* slots = has_valid ? ctx->scratch : nullptr;  // zero between calls
  ScatterArgs a{};
  a.values = values->values;
  a.mask = p->mask;
  a.mask_valid = p->mask_valid;
  a.vvalid = has_valid ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  a.len = p->len;
  a.chunk_prefix = p->chunk_prefix;
  a.group_prefix = p->group_prefix;
  a.group_shift = p->group_shift;
  a.out_values = dst_values;
  a.out_valid = (unsigned long long*)dst_validity;
  a.valid_slots = slots;
  a.out_base = dst_row_offset;
  const bool skip = use_skip(K, p->len);
  {
    ah_prof_scope ps(ctx, "filter_scatter");
    if (has_valid) launch_scatter<true>(ctx, width, a, skip, 1, use_sparse(K, p->len), K);
    else launch_scatter<false>(ctx, width, a, skip, 1, use_sparse(K, p->len), K);
  }
  hipError_t e = hipGetLastError();
  ah_status st = AH_OK;
  if (e == hipSuccess && !has_valid)  // source without nulls: the appended rows are all valid
    st = ah_bitmap_set_bits(ctx, dst_validity, dst_row_offset, nullptr, 0, K, nullptr);
  if (no_wait) {  // the null count accumulates on the device; the caller reads it when the batch is finished
    ctx->inflight = true;
    if (e == hipSuccess && has_valid) {
      filter_finish_acc_kernel<<<1, 64, 0, ctx->stream>>>(slots, (unsigned long long)K, nulls_acc);
      e = hipGetLastError();
    }
    if (st != AH_OK) return st;
    if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
    return AH_OK;
  }
  if (e == hipSuccess && has_valid) {
    const uint64_t seq = ah_mail_next(ctx);
    filter_finish_kernel<<<1, 64, 0, ctx->stream>>>(slots, ctx->pinned_dev, seq);
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  } else if (e == hipSuccess) {
    e = ah_stream_wait(ctx);
  }
  if (st != AH_OK) return st;
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
  if (has_valid && appended_nulls) *appended_nulls = K - (int64_t)ctx->pinned[0];
  return AH_OK;
}

// The coalescer's scatters for sparse selections (W <= 8): m.ntiles counted in 4096-row tiles; the launch appends
// `appended` rows at destination row `dest0` of every column; their NULL rows are counted behind it into m.dst[c].null_slots.
static ah_status launch_multi_sparse(ah_context* ctx, const MultiArgs& m, int width, int ncols, int64_t dest0, int64_t appended) {
  const int64_t nwg = (m.ntiles + 3) >> 2;
  const dim3 grid((unsigned)(m.xcd_remap ? 8 * ((nwg + 7) / 8) : nwg), (unsigned)ncols);
  switch (width) {
    case 1: filter_scatter_multi_sparse_kernel<1><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
    case 2: filter_scatter_multi_sparse_kernel<2><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
    case 4: filter_scatter_multi_sparse_kernel<4><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
    case 8: filter_scatter_multi_sparse_kernel<8><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
    default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
  }
  if (appended > 0) {
    RangeCount r{};
    for (int c = 0; c < ncols; ++c) r.bits[c] = m.dst[c].out_valid, r.slots[c] = m.dst[c].null_slots;
    r.bit_lo = dest0, r.nbits = appended;
    const int64_t nwords = ((dest0 + appended - 1) >> 6) - (dest0 >> 6) + 1;
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ah_ceil_div(nwords, 256 * 4)));
    range_null_count_kernel<<<dim3(gx, (unsigned)ncols), 256, 0, ctx->stream>>>(r);
  }
  return AH_OK;
}

// ---- batch tables (filter_internal.hpp)
ah_status ah_filter_table_count(ah_context* ctx, const ah_tbl_push& t, uint64_t* pin_dev, uint64_t* seq_out) {
  if (t.nwaves < 1 || t.nsegs < 1) return AH_INVALID_ARGUMENT;
  ctx->inflight = true;
  const uint64_t seq = ah_mail_next(ctx);
  *seq_out = seq;
  {
    ah_prof_scope ps(ctx, "filter_count");
    filter_count_table_kernel<<<(unsigned)t.nwaves, 64, 0, ctx->stream>>>(t);
    filter_scan_table_kernel<<<1, 1024, 0, ctx->stream>>>(t, pin_dev, ctx->pinned_dev, seq);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "filter count failed: %s", hipGetErrorString(e));
  return AH_OK;
}

ah_status ah_filter_table_scatter(ah_context* ctx, const ah_tbl_push& t, int width, int ncols, const int* col_index,
                                  const ah_tbl_dst* dst, int64_t tile_lo, int64_t tile_hi, int64_t win_lo, int64_t win_hi,
                                  int64_t out_base, bool aligned16, bool sparse, bool skip) {
  if (ncols < 1 || ncols > AH_TBL_MAX_COLS || tile_hi <= tile_lo || win_hi <= win_lo) return ncols < 1 || ncols > AH_TBL_MAX_COLS ? AH_INVALID_ARGUMENT : AH_OK;
  TblScatterArgs m{};
  m.t = t;
  m.tile_lo = tile_lo;
  m.ntiles = tile_hi - tile_lo;
  static const char* xr = getenv("AH_FILTER_XCD");
  m.xcd_remap = (xr && xr[0] == '0') ? 0 : 1;
  m.win_lo = win_lo, m.win_hi = win_hi, m.out_base = out_base;
  for (int c = 0; c < ncols; ++c) m.col_index[c] = col_index[c], m.dst[c] = dst[c];
  ctx->inflight = true;
  ah_prof_scope ps(ctx, "filter_scatter");
  if (sparse) {
    const int64_t nwg = (m.ntiles + 3) >> 2;
    const dim3 grid((unsigned)(m.xcd_remap ? 8 * ((nwg + 7) / 8) : nwg), (unsigned)ncols);
    switch (width) {
      case 1: filter_scatter_table_sparse_kernel<1><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
      case 2: filter_scatter_table_sparse_kernel<2><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
      case 4: filter_scatter_table_sparse_kernel<4><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
      case 8: filter_scatter_table_sparse_kernel<8><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m); break;
      default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
    }
  } else {
    const dim3 grid((unsigned)(m.xcd_remap ? 8 * ((m.ntiles + 7) / 8) : m.ntiles), (unsigned)ncols);
#define AH_TBL(W, V)                                                                                        \
  if (!aligned16) filter_scatter_table_kernel<W, 1, false><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m);   \
  else if (skip) filter_scatter_table_kernel<W, V, true><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m);     \
  else filter_scatter_table_kernel<W, V, false><<<grid, SCATTER_THREADS, 0, ctx->stream>>>(m)
    switch (width) {
      case 1: AH_TBL(1, 16); break;
      case 2: AH_TBL(2, 8); break;
      case 4: AH_TBL(4, 4); break;
      case 8: AH_TBL(8, 2); break;
      default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
    }
#undef AH_TBL
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "table scatter failed: %s", hipGetErrorString(e));
  return AH_OK;
}

ah_status ah_filter_count_nulls_batches(ah_context* ctx, int ncols, const unsigned long long* const* bits, int64_t stride,
                                        int64_t nbatches, int64_t last_rows, unsigned long long* out, uint64_t* seq_out) {
  if (ncols < 1 || ncols > AH_TBL_MAX_COLS || (stride & 63) || nbatches < 1) return AH_INVALID_ARGUMENT;
  NullBatches nb{};
  for (int c = 0; c < ncols; ++c) nb.bits[c] = bits[c];
  const int64_t nwords = stride >> 6, n = nbatches * ncols;
  nb.parts = (int)std::max<int64_t>(1, std::min<int64_t>(64, ah_ceil_div(nwords, 256 * 8)));
  nb.ncols = ncols, nb.stride = stride, nb.nbatches = nbatches, nb.last_rows = last_rows;
  void* acc = nullptr;
  AH_TRY(ah_pool_alloc(ctx, (size_t)n * 8, &acc));
  nb.acc = (unsigned long long*)acc;
  hipMemsetAsync(acc, 0, (size_t)n * 8, ctx->stream);
  count_nulls_batches_kernel<<<dim3((unsigned)(nbatches * nb.parts), (unsigned)ncols), 256, 0, ctx->stream>>>(nb);
  const uint64_t seq = ah_mail_next(ctx);
  *seq_out = seq;
  words_to_pinned_kernel<<<1, 1024, 0, ctx->stream>>>(nb.acc, n, out, ctx->pinned_dev, seq);
  ah_pool_free(ctx, acc);  // (reuse is stream-ordered behind the two kernels)
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "null count failed: %s", hipGetErrorString(e));
  return AH_OK;
}
ah_status ah_filter_count_nulls_range(ah_context* ctx, int ncols, const unsigned long long* const* bits, unsigned long long* const* slots,
                                      int64_t bit_lo, int64_t nbits) {
  if (ncols < 1 || ncols > SCATTER_MAX_COLS) return AH_INVALID_ARGUMENT;
  if (nbits <= 0) return AH_OK;
  RangeCount r{};
  for (int c = 0; c < ncols; ++c) r.bits[c] = bits[c], r.slots[c] = slots[c];
  r.bit_lo = bit_lo, r.nbits = nbits;
  const int64_t nwords = ((bit_lo + nbits - 1) >> 6) - (bit_lo >> 6) + 1;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ah_ceil_div(nwords, 256 * 4)));
  range_null_count_kernel<<<dim3(gx, (unsigned)ncols), 256, 0, ctx->stream>>>(r);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "null count failed: %s", hipGetErrorString(e));
  return AH_OK;
}

// Several columns of one batch through ONE scatter launch (BatchCoalescer's filtered push): the columns must share
// the value width and all carry validity; returns AH_NOT_YET_IMPLEMENTED (nothing enqueued) when they do not, and the
// caller goes column by column.  Only positions [win_lo, win_hi) of the filtered batch are appended (at
// dst_row_offset onwards): a batch that straddles two output batches is two launches, no intermediate array.
// `null_slots`: ncols x 64 device words that ACCUMULATE the NULL rows appended (each tile adds its own window rows -
// valid rows): no tail kernel, nothing to read back per push — the coalescer sums them once per finished batch.
ah_status ah_filter_apply_into_acc_cols(ah_context* ctx, const ah_filter_predicate* p, int ncols, const ah_array_view* values,
                                        void* const* dst_values, uint8_t* const* dst_validity, int64_t dst_row_offset,
                                        unsigned long long* null_slots, int64_t win_lo, int64_t win_hi, int speculative,
                                        double selectivity_hint) {
  if (ncols < 1 || ncols > SCATTER_MAX_COLS) return AH_NOT_YET_IMPLEMENTED;
  const int width = ah_type_width(values[0].type);
  if (width <= 0) return AH_NOT_YET_IMPLEMENTED;
  for (int c = 0; c < ncols; ++c) {
    if (ah_type_width(values[c].type) != width || !values[c].validity || values[c].null_count == 0) return AH_NOT_YET_IMPLEMENTED;
    if (p->len > values[c].length || ((uintptr_t)dst_validity[c] & 7) != 0) return AH_NOT_YET_IMPLEMENTED;
  }
  // speculative: the count pass is enqueued but K has not reached the host; the launch clips itself to the rows
  // that exist (a tile outside the window or without selected rows exits)
  if (!speculative && win_hi > p->count) win_hi = p->count;
  const int64_t K = win_hi - win_lo;  // rows this launch appends (at most, when speculative): positions [win_lo, win_hi)
  if (p->len == 0 || K <= 0) return AH_OK;
  if (!speculative && width <= 8 && use_sparse(p->count, p->len)) {  // a sparse selection: a tile per wave, NULLs counted after
    MultiArgs m{};
    m.nsegs = 1;
    static const char* xr = getenv("AH_FILTER_XCD");
    m.xcd_remap = (xr && xr[0] == '0') ? 0 : 1;
    MultiSeg& sg = m.seg[0];
    sg.mask = p->mask, sg.mask_valid = p->mask_valid, sg.len = p->len;
    sg.chunk_prefix = p->chunk_prefix, sg.group_prefix = p->group_prefix, sg.group_shift = p->group_shift;
    sg.out_base = dst_row_offset, sg.win_lo = win_lo, sg.win_hi = win_hi, sg.tile0 = 0;
    m.ntiles = ah_ceil_div(p->len, 4096);
    for (int c = 0; c < ncols; ++c) {
      sg.col[c].values = values[c].values;
      sg.col[c].vvalid = make_bitview(values[c].validity, values[c].validity_bit_offset);
      m.dst[c].out_values = dst_values[c];
      m.dst[c].out_valid = (unsigned long long*)dst_validity[c];
      m.dst[c].null_slots = null_slots + (size_t)c * 64;
    }
    ctx->inflight = true;
    ah_status st;
    {
      ah_prof_scope ps(ctx, "filter_scatter");
      st = launch_multi_sparse(ctx, m, width, ncols, dst_row_offset, K);
    }
    const hipError_t e = hipGetLastError();
    if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
    return st;
  }
  ScatterArgs a{};
  a.mask = p->mask;
  a.mask_valid = p->mask_valid;
  a.len = p->len;
  a.chunk_prefix = p->chunk_prefix;
  a.group_prefix = p->group_prefix;
  a.group_shift = p->group_shift;
  a.out_base = dst_row_offset;
  a.nulls_mode = 1;
  // (always windowed here: nulls_mode needs each tile's clipped row count, which the window arithmetic provides)
  a.win_lo = win_lo, a.win_hi = win_hi;
  a.values = values[0].values;
  a.vvalid = make_bitview(values[0].validity, values[0].validity_bit_offset);
  a.out_values = dst_values[0];
  a.out_valid = (unsigned long long*)dst_validity[0];
  a.valid_slots = null_slots;
  for (int c = 1; c < ncols; ++c) {
    ScatterArgs::Col& m = a.more[c - 1];
    m.values = values[c].values;
    m.vvalid = make_bitview(values[c].validity, values[c].validity_bit_offset);
    m.out_values = dst_values[c];
    m.out_valid = (unsigned long long*)dst_validity[c];
    m.valid_slots = null_slots + (size_t)c * 64;
  }
  ctx->inflight = true;
  {
    ah_prof_scope ps(ctx, "filter_scatter");
    const bool skip = speculative ? use_skip((int64_t)(selectivity_hint * (double)p->len), p->len) : use_skip(p->count, p->len);
    launch_scatter<true>(ctx, width, a, skip, ncols);  // (windowed, NULL-counting: the tiled kernel)
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
  return AH_OK;
}

// Tiles [*t_lo, *t_hi) of predicate p that can hold positions [lo, hi) of its filtered stream (hi == 0: to the end), from
// the predicate's quantile prefixes; everything when it has none (window_tiles.hpp).  A batch cut by an output-batch
// boundary is two segments: without the bound each walks all of the batch's tiles and exits early from the ones outside
// its window — 8192 workgroups per cut that only cost dispatch (~25 us per cut, 0.35 ms per 1e9 rows at four batches per
// output batch).
static void window_tiles(const ah_filter_predicate* p, int64_t lo, int64_t hi, int T, int64_t* t_lo, int64_t* t_hi) {
  static const char* off = getenv("AH_FILTER_WINDOW_TILES");  // "0": walk every tile (A/B runs)
  if (p->quant_n <= 0 || (off && off[0] == '0')) return;
  const int64_t rows_per_group = ((int64_t)1 << p->group_shift) * AH_FILTER_CHUNK_ROWS;  // (a multiple of every tile size)
  ah_window_tiles(p->quant_n, p->quant_step, p->quant, p->count, rows_per_group, lo, hi, T, *t_hi, t_lo, t_hi);
}

// Up to 8 (batch, window) segments through ONE launch, all appending to the same in-progress columns (the grouped
// push of BatchCoalescer).  Preconditions (the caller checks them like ah_filter_apply_into_acc_cols does): every
// column of every batch has the same width and carries validity; dst_validity 8-byte aligned.
ah_status ah_filter_apply_multi(ah_context* ctx, int nsegs, const ah_filter_predicate* const* preds,
                                const ah_array_view* const* columns, const int64_t* win_lo, const int64_t* win_hi,
                                const int64_t* out_base, int ncols, void* const* dst_values, uint8_t* const* dst_validity,
                                unsigned long long* null_slots) {
  if (nsegs < 1 || nsegs > MULTI_MAX_SEGS || ncols < 1 || ncols > SCATTER_MAX_COLS) return AH_INVALID_ARGUMENT;
  const int width = ah_type_width(columns[0][0].type);
  if (width <= 0)  // Boolean / string / view columns have no fixed-width in-progress buffer to scatter into
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "grouped filter scatter of %s columns", ah_type_name(columns[0][0].type));
  for (int c = 0; c < ncols; ++c)
    if (!dst_values[c] || !dst_validity[c]) return ah_fail(ctx, AH_INVALID_ARGUMENT, "grouped filter scatter: column %d has no destination", c);
  MultiArgs m{};
  m.nsegs = nsegs;
  static const char* xr = getenv("AH_FILTER_XCD");
  m.xcd_remap = (xr && xr[0] == '0') ? 0 : 1;
  const int T = tile_rows(width);
  int64_t tiles = 0, rows = 0, selected = 0;
  bool aligned16 = true;
  for (int i = 0; i < nsegs; ++i) {
    const ah_filter_predicate* p = preds[i];
    MultiSeg& sg = m.seg[i];
    sg.mask = p->mask;
    sg.mask_valid = p->mask_valid;
    sg.len = p->len;
    sg.chunk_prefix = p->chunk_prefix;
    sg.group_prefix = p->group_prefix;
    sg.group_shift = p->group_shift;
    sg.out_base = out_base[i];
    sg.win_lo = win_lo[i];
    sg.win_hi = win_hi[i];
    sg.tile0 = tiles;
    int64_t t_lo = 0, t_hi = ah_ceil_div(p->len, T);
    window_tiles(p, win_lo[i], win_hi[i], T, &t_lo, &t_hi);
    sg.tile_lo = t_lo;
    tiles += t_hi - t_lo;
    rows += p->len;
    selected += p->count;
    for (int c = 0; c < ncols; ++c) {
      sg.col[c].values = columns[i][c].values;
      sg.col[c].vvalid = make_bitview(columns[i][c].validity, columns[i][c].validity_bit_offset);
      aligned16 = aligned16 && (((uintptr_t)columns[i][c].values) & 15) == 0;
    }
  }
  m.ntiles = tiles;
  for (int c = 0; c < ncols; ++c) {
    m.dst[c].out_values = dst_values[c];
    m.dst[c].out_valid = (unsigned long long*)dst_validity[c];
    m.dst[c].null_slots = null_slots + (size_t)c * 64;
  }
  if (tiles == 0) return AH_OK;
  ctx->inflight = true;
  if (width <= 8 && use_sparse(selected, rows)) {  // (tile_rows(width <= 8) == 4096: `tiles` / tile0 already are in wave tiles)
    // rows this launch appends: consecutive destination rows from the first segment's out_base on
    int64_t appended = 0;
    for (int i = 0; i < nsegs; ++i) {
      const int64_t hi = (win_hi[i] != 0 && win_hi[i] < preds[i]->count) ? win_hi[i] : preds[i]->count;
      if (hi > win_lo[i]) appended += hi - win_lo[i];
    }
    ah_status st;
    {
      ah_prof_scope ps(ctx, "filter_scatter");
      st = launch_multi_sparse(ctx, m, width, ncols, out_base[0], appended);
    }
    const hipError_t e = hipGetLastError();
    if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
    return st;
  }
  const bool skip = use_skip(selected, rows);
  const dim3 grid((unsigned)(m.xcd_remap ? 8 * ((tiles + 7) / 8) : tiles), (unsigned)ncols), block(SCATTER_THREADS);
  {
    ah_prof_scope ps(ctx, "filter_scatter");
#define AH_MULTI(W, V)                                                                                  \
  do {                                                                                                  \
    if (skip) filter_scatter_multi_kernel<W, V, true><<<grid, block, 0, ctx->stream>>>(m);             \
    else filter_scatter_multi_kernel<W, V, false><<<grid, block, 0, ctx->stream>>>(m);                 \
  } while (0)
    switch (width) {
      case 1: if (aligned16) AH_MULTI(1, 16); else AH_MULTI(1, 1); break;
      case 2: if (aligned16) AH_MULTI(2, 8); else AH_MULTI(2, 1); break;
      case 4: if (aligned16) AH_MULTI(4, 4); else AH_MULTI(4, 1); break;
      case 8: if (aligned16) AH_MULTI(8, 2); else AH_MULTI(8, 1); break;
      case 16: AH_MULTI(16, 1); break;
      case 32: AH_MULTI(32, 1); break;
      default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
    }
#undef AH_MULTI
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "fused filter copy failed: %s", hipGetErrorString(e));
  return AH_OK;
}

// the per-column NULL-row counters of a BatchCoalescer (ncols x 64 slots) -> the coalescer's own pinned words `dst`
// (device view of host memory), counters back to zero, then the context mailbox is posted with `seq`: ONE launch per
// finished output batch and NO wait — the host reads the counts when the batch is fetched (ah_mail_wait(seq), long
// passed by then).  NullBufferBuilder only needs the count at finish (coalesce/primitive.rs:94-106).
__global__ void __launch_bounds__(64) coalesce_finish_kernel(unsigned long long* slots, int ncols, uint64_t* dst, uint64_t* mail,
                                                             uint64_t seq) {
  for (int c = 0; c < ncols; ++c) {
    unsigned long long v = slots[(size_t)c * 64 + threadIdx.x];
    slots[(size_t)c * 64 + threadIdx.x] = 0;
    v = wave_reduce_add64(v);
    if (threadIdx.x == 0) __hip_atomic_store(dst + c, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) ah_mail_post(mail, seq);
}
ah_status ah_coalesce_post_nulls(ah_context* ctx, unsigned long long* slots, int ncols, uint64_t* pinned_dst_dev, uint64_t* seq_out) {
  if (ncols < 1 || ncols > 200) return AH_INVALID_ARGUMENT;
  const uint64_t seq = ah_mail_next(ctx);
  coalesce_finish_kernel<<<1, 64, 0, ctx->stream>>>(slots, ncols, pinned_dst_dev, ctx->pinned_dev, seq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "coalescer finish failed: %s", hipGetErrorString(e));
  *seq_out = seq;
  return AH_OK;
}
ah_status ah_coalesce_wait(ah_context* ctx, uint64_t seq) {
  hipError_t e = ah_mail_wait(ctx, seq);
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "coalescer finish failed: %s", hipGetErrorString(e));
  return AH_OK;
}

extern "C" ah_status ah_filter_predicate_apply_into(ah_context* ctx, const ah_filter_predicate* p,
                                                    const ah_array_view* values, void* dst_values,
                                                    uint8_t* dst_validity, int64_t dst_row_offset,
                                                    int64_t* appended_nulls) {
  ah_ctx_guard _guard(ctx);
  return apply_into_impl(ctx, p, values, dst_values, dst_validity, dst_row_offset, appended_nulls, nullptr, false);
}

// The same without a host wait (BatchCoalescer's hot loop, arrow-select/src/coalesce.rs:229): the number of NULL rows
// appended is added to the device word *nulls_acc, read once when the in-progress batch is finished (ah_read_words).
extern "C" ah_status ah_filter_predicate_apply_into_acc(ah_context* ctx, const ah_filter_predicate* p,
                                                        const ah_array_view* values, void* dst_values,
                                                        uint8_t* dst_validity, int64_t dst_row_offset,
                                                        uint64_t* nulls_acc) {
  ah_ctx_guard _guard(ctx);
  if (!nulls_acc) return AH_INVALID_ARGUMENT;
  return apply_into_impl(ctx, p, values, dst_values, dst_validity, dst_row_offset, nullptr, (unsigned long long*)nulls_acc, true);
}

// AH_FILTER_SMALL=0 keeps small inputs on the general two-pass path (tests run both; read per call: a getenv is ~0.1 us)
static bool small_path_enabled() {
  const char* e = getenv("AH_FILTER_SMALL");
  return !(e && e[0] == '0');
}

extern "C" ah_status ah_filter(ah_context* ctx, const ah_array_view* values,
                               const ah_array_view* predicate, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !predicate || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  if (predicate->type == AH_BOOL && predicate->length > values->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT,
                   "Filter predicate of length %lld is larger than target array of length %lld",
                   (long long)predicate->length, (long long)values->length);
  if (small_path_enabled()) {  // query-engine-sized batches: one launch, one wait (filter_small.hip)
    const ah_status ss = ah_filter_small(ctx, 1, values, predicate, out, nullptr);
    if (ss != AH_NOT_YET_IMPLEMENTED) return ss;
  }
  ah_filter_predicate* p = nullptr;
  AH_TRY(ah_filter_predicate_build(ctx, predicate, &p));
  ah_status st = ah_filter_predicate_apply(ctx, p, values, out);
  ah_filter_predicate_free(ctx, p);
  return st;
}

// filter_record_batch's primitive columns of one width and validity shape through ONE scatter launch and ONE host
// wait (`cols`: indices into columns / outs, 2..8 of them).  The per-column path costs a launch pair and a wait per
// column, which is what a query engine's 8 Ki-row batches of many columns pay for.
static ah_status filter_columns_fused(ah_context* ctx, const ah_filter_predicate* p, const ah_array_view* columns,
                                      ah_array_out* outs, const int* cols, int ncols, int width, bool has_valid) {
  const int64_t K = p->count;
  const size_t vbytes = (size_t)K * width, bbytes = has_valid ? ah_bitmap_bytes(K) : 0;
  void* ov[SCATTER_MAX_COLS] = {};
  void* ob[SCATTER_MAX_COLS] = {};
  ah_status st = AH_OK;
  for (int i = 0; i < ncols && st == AH_OK; ++i) {
    st = ah_out_alloc(ctx, vbytes, &ov[i]);
    if (st == AH_OK && has_valid) st = ah_out_alloc(ctx, bbytes, &ob[i]);
  }
  if (st == AH_OK && has_valid) {
    ZeroCols z{};
    for (int i = 0; i < ncols; ++i) z.p[i] = (unsigned long long*)ob[i];
    const int64_t words = (int64_t)(bbytes / 8);  // ah_bitmap_bytes: whole 64-bit words
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(words, 256), 2048));
    zero_words_cols_kernel<<<dim3(gx, (unsigned)ncols), 256, 0, ctx->stream>>>(z, words);
  }
  hipError_t e = hipSuccess;
  if (st == AH_OK) {
    ScatterArgs a{};
    a.mask = p->mask;
    a.mask_valid = p->mask_valid;
    a.len = p->len;
    a.chunk_prefix = p->chunk_prefix;
    a.group_prefix = p->group_prefix;
      a.group_shift = p->group_shift;
    for (int i = 0; i < ncols; ++i) {
      const ah_array_view& v = columns[cols[i]];
      const BitView vv = has_valid ? make_bitview(v.validity, v.validity_bit_offset) : BitView{nullptr, 0};
      if (i == 0) {
        a.values = v.values, a.vvalid = vv, a.out_values = ov[0], a.out_valid = (unsigned long long*)ob[0];
        a.valid_slots = has_valid ? ctx->scratch : nullptr;
      } else {
        ScatterArgs::Col& m = a.more[i - 1];
        m.values = v.values, m.vvalid = vv, m.out_values = ov[i], m.out_valid = (unsigned long long*)ob[i];
        m.valid_slots = has_valid ? ctx->scratch + (size_t)i * 64 : nullptr;
      }
    }
    {
      ah_prof_scope ps(ctx, "filter_scatter");
      if (has_valid) st = launch_scatter<true>(ctx, width, a, use_skip(K, p->len), ncols, use_sparse(K, p->len), K);
      else st = launch_scatter<false>(ctx, width, a, use_skip(K, p->len), ncols, use_sparse(K, p->len), K);
    }
    e = hipGetLastError();
    if (st == AH_OK && e == hipSuccess) {
      if (has_valid) {
        const uint64_t seq = ah_mail_next(ctx);
        filter_finish_cols_kernel<<<1, 64, 0, ctx->stream>>>(ctx->scratch, ncols, ctx->pinned_dev, seq);
        e = hipGetLastError();
        if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
      } else {
        e = ah_stream_wait(ctx);
      }
    }
    if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "filter scatter failed: %s", hipGetErrorString(e));
  }
  if (st != AH_OK) {
    for (int i = 0; i < ncols; ++i) {
      ah_out_free(ctx, ov[i], vbytes);
      ah_out_free(ctx, ob[i], bbytes);
    }
    return st;
  }
  for (int i = 0; i < ncols; ++i) {
    ah_array_out* out = &outs[cols[i]];
    out->type = columns[cols[i]].type;
    out->length = K;
    out->values = ov[i];
    out->values_bytes = (int64_t)vbytes;
    if (has_valid) {
      const int64_t nulls = K - (int64_t)ctx->pinned[i];
      if (nulls == 0) {  // filter_nulls :523-525 -> None
        ah_out_free(ctx, ob[i], bbytes);
      } else {
        out->validity = (uint8_t*)ob[i];
        out->validity_bytes = (int64_t)bbytes;
        out->null_count = nulls;
      }
    }
  }
  return AH_OK;
}

extern "C" ah_status ah_filter_record_batch(ah_context* ctx, int32_t n_columns,
                                            const ah_array_view* columns,
                                            const ah_array_view* predicate, ah_array_out* outs,
                                            int64_t* out_rows) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !predicate || (n_columns > 0 && (!columns || !outs))) return AH_INVALID_ARGUMENT;
  for (int32_t c = 0; c < n_columns; ++c) ah_out_init(&outs[c]);
  if (n_columns > 0 && small_path_enabled()) {  // all columns in one launch per shape, one wait for the whole batch
    const ah_status ss = ah_filter_small(ctx, n_columns, columns, predicate, outs, out_rows);
    if (ss != AH_NOT_YET_IMPLEMENTED) {
      if (ss != AH_OK)
        for (int32_t c = 0; c < n_columns; ++c) ah_out_init(&outs[c]);
      return ss;
    }
  }
  ah_filter_predicate* p = nullptr;
  AH_TRY(ah_filter_predicate_build(ctx, predicate, &p));
  ah_status st = AH_OK;
  // same-shape primitive columns go out together; everything else (and the None / All strategies, deferred mode,
  // error cases) column by column
  std::vector<char> done((size_t)std::max(n_columns, 1), 0);
  if (!ctx->deferred && p->len > 0 && p->count > 0 && p->count < p->len) {
    std::vector<int> width((size_t)n_columns, 0);
    std::vector<char> hv((size_t)n_columns, 0);
    for (int32_t c = 0; c < n_columns && st == AH_OK; ++c) {
      const ah_array_view& v = columns[c];
      const bool prim = v.type != AH_BOOL && v.type != AH_UTF8 && v.type != AH_LARGE_UTF8 && ah_type_width(v.type) > 0;
      if (!prim || p->len > v.length) continue;
      int64_t in_nulls = 0;
      st = ah_resolve_null_count(ctx, &v, &in_nulls);
      width[c] = ah_type_width(v.type);
      hv[c] = v.validity && in_nulls > 0;
    }
    for (int32_t c = 0; c < n_columns && st == AH_OK; ++c) {
      if (done[c] || width[c] == 0) continue;
      int group[SCATTER_MAX_COLS], g = 0;
      for (int32_t d = c; d < n_columns && g < SCATTER_MAX_COLS; ++d)
        if (!done[d] && width[d] == width[c] && hv[d] == hv[c]) group[g++] = d;
      if (g < 2) continue;
      st = filter_columns_fused(ctx, p, columns, outs, group, g, width[c], hv[c] != 0);
      for (int i = 0; i < g; ++i) done[group[i]] = 1;
    }
  }
  for (int32_t c = 0; c < n_columns && st == AH_OK; ++c)
    if (!done[c]) st = ah_filter_predicate_apply(ctx, p, &columns[c], &outs[c]);
  if (st != AH_OK)
    for (int32_t c = 0; c < n_columns; ++c) ah_array_release(ctx, &outs[c]);
  if (out_rows) *out_rows = p->count;  // RecordBatch row_count = predicate.count (filter.rs:476)
  ah_filter_predicate_free(ctx, p);
  return st;
}
