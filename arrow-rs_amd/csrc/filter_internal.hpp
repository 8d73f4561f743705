// filter_internal.hpp — what filter.hip shares with filter_expr.hip (the lazily evaluated predicate) and coalesce.hip.
#pragma once
#include "common.hpp"

#ifdef __HIPCC__
// FilterPredicate (filter.rs:442-449): predicate bits (borrowed, or owned when built from an expression), count and
// the device-resident prefix tables that replace IterationStrategy::Indices.
constexpr int AH_FILTER_QUANTS = 32;
struct ah_filter_predicate {
  BitView mask, mask_valid;
  int64_t len = 0;
  int64_t count = 0;
  uint32_t* chunk_prefix = nullptr;
  unsigned long long* group_prefix = nullptr;
  int group_shift = 10;
  void* block = nullptr;  // single pool allocation backing the tables (and the mask of an expression predicate)
  unsigned long long* total_dev = nullptr;  // K on the device (kernels that run before the host has read it)
  // Coarse host-side view of WHERE the selected rows are (predicates counted through ah_filter_predicates_begin with a
  // quantile area): quant[k] = selected rows in count groups [0, k * quant_step), k = 0 .. quant_n, quant[quant_n] = count.
  // A caller that appends only positions [lo, hi) of the filtered stream (BatchCoalescer: a batch cut by an output-batch
  // boundary) bounds the tiles worth launching with it instead of walking — and early-exiting from — every tile.
  int quant_n = 0;  // 0: not recorded
  int64_t quant_step = 0;
  uint64_t quant[AH_FILTER_QUANTS + 1] = {};
};

constexpr int AH_FILTER_CHUNK_ROWS = 1024;

// rows per scatter tile / LDS staging capacity in elements (16 KiB of LDS per workgroup => 8 workgroups / CU)
__host__ __device__ constexpr int tile_rows(int width) { return width <= 8 ? 4096 : (width == 16 ? 2048 : 1024); }
__host__ __device__ constexpr int stage_cap(int width) { return width <= 8 ? 2048 : (width == 16 ? 1024 : 512); }

template <int W> struct Elem;
template <> struct Elem<1> { using type = uint8_t; };
template <> struct Elem<2> { using type = uint16_t; };
template <> struct Elem<4> { using type = uint32_t; };
template <> struct Elem<8> { using type = uint64_t; };
// 16- and 32-byte natives as clang vector types: first-class values the optimizer keeps in registers (as structs of
// four dwords the per-thread arrays of them stayed in scratch: 144 bytes per thread in the Decimal128 / i256 variants)
typedef uint32_t E16 __attribute__((ext_vector_type(4)));
typedef uint32_t E32 __attribute__((ext_vector_type(8), aligned(16)));  // i256 buffers are only 16-byte aligned
template <> struct Elem<16> { using type = E16; };
template <> struct Elem<32> { using type = E32; };

template <int W, int V> struct alignas((W * V >= 16) ? 16 : W * V) Vec {
  using T = typename Elem<W>::type;
  union { T e[V]; } u;
};

// filter_small.hip: the one-launch path for predicates of at most AH_FILTER_SMALL_MAX rows.  Returns AH_NOT_YET_IMPLEMENTED
// (nothing enqueued) when a column is not a fixed-width primitive; the caller takes the general path.
constexpr int64_t AH_FILTER_SMALL_MAX = 1 << 20;
ah_status ah_filter_small(ah_context* ctx, int ncols, const ah_array_view* columns, const ah_array_view* predicate,
                          ah_array_out* outs, int64_t* out_rows);  // granule of the count pass: 16 mask words

// ---- batch tables in device memory (round 5; BatchCoalescer's slab push at the reference's operating point: thousands of
// 8192-row input batches per call, coalesce.rs:172-173).  The grouped push of round 4 passed up to 8 (batch, window) segments
// BY VALUE in the kernel arguments: 1 900 grouped calls and 12 000 output-batch allocations per 1e9 rows at 8192-row batches.
// Here the host writes one table entry per batch (and per 1024-row chunk, and per 4096-row tile), uploads the tables
// with one copy, and ONE count launch + ONE scan + ONE scatter launch per destination cover the whole push.
constexpr int AH_TBL_MAX_COLS = 8;
struct ah_tbl_seg {   // one input batch
  BitView mask, mask_valid;   // mask_valid.words == nullptr: the predicate has no nulls
  int64_t len;                // predicate length (<= the batch's rows)
  int64_t chunk0;             // this batch's first chunk (1024 rows) in the push's global chunk numbering
};
struct ah_tbl_col {   // column k of batch i: entry i * ncols + k
  const void* values;
  BitView vvalid;             // words == nullptr: every row valid
};
struct ah_tbl_tile { int32_t seg, tile; };    // scatter tile t: 4096-row tile `tile` of batch `seg`
struct ah_tbl_push {                          // all device pointers
  const ah_tbl_seg* segs;
  const ah_tbl_col* cols;
  int ncols;
  const int32_t* chunk_seg;          // [nchunks]: the batch of each global chunk
  const ah_tbl_tile* tiles;
  int64_t nsegs, nchunks, nwaves, ntiles;  // nwaves = ceil(nchunks / 64): a count wave owns 64 consecutive global chunks
  uint32_t* chunk_prefix;            // [nchunks]: selected rows of the chunk's wave before the chunk
  uint32_t* wave_total;              // selected rows per count wave
  unsigned long long* wave_prefix;   // [nwaves + 1]: position of each wave's first selected row in the push's filtered stream
};
// count + scan, enqueued only: pin_dev[w] = position of count wave w's first selected row (w = 0 .. nwaves - 1), pin_dev[nwaves] = K
// ... and *seq = the mailbox sequence number the scan posts behind them (ah_mail_wait)
ah_status ah_filter_table_count(ah_context* ctx, const ah_tbl_push& t, uint64_t* pin_dev, uint64_t* seq);
struct ah_tbl_dst {
  void* out_values;
  unsigned long long* out_valid;  // zero-initialised words
};
// positions [win_lo, win_hi) of the push's filtered stream -> destination rows out_base + (pos - win_lo), for `ncols` columns
// of one value width (launch column y reads table column col_index[y]); tiles [tile_lo, tile_hi) of the tile table
ah_status ah_filter_table_scatter(ah_context* ctx, const ah_tbl_push& t, int width, int ncols, const int* col_index,
                                  const ah_tbl_dst* dst, int64_t tile_lo, int64_t tile_hi, int64_t win_lo, int64_t win_hi,
                                  int64_t out_base, bool aligned16, bool sparse, bool skip);
// NULL rows (0 bits) of bit ranges, counted after a scatter that kept no counters:
//   batch form: for j < nbatches, column c < ncols: out[j * ncols + c] = zero bits among bits [j * stride, j * stride + rows_j)
//   of bits[c] (rows_j = stride except for the last batch: last_rows); `out` may be device-visible pinned memory
ah_status ah_filter_count_nulls_batches(ah_context* ctx, int ncols, const unsigned long long* const* bits, int64_t stride,
                                        int64_t nbatches, int64_t last_rows, unsigned long long* out, uint64_t* seq);
//   range form: slots[c][..] += zero bits among bits [bit_lo, bit_lo + nbits) of bits[c]   (c < ncols <= 8)
ah_status ah_filter_count_nulls_range(ah_context* ctx, int ncols, const unsigned long long* const* bits, unsigned long long* const* slots,
                                      int64_t bit_lo, int64_t nbits);

// filter.hip: K2 (scan of the group totals; K lands in pinned slot `slot`, the mailbox is posted with `seq` != 0)
void ah_filter_launch_group_scan(ah_context* ctx, const uint32_t* group_total, int64_t ngroups, unsigned long long* group_prefix,
                                 unsigned long long* total, int slot, uint64_t seq);
#endif
