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

// filter.hip: K2 (scan of the group totals; K lands in pinned slot `slot`, the mailbox is posted with `seq` != 0)
void ah_filter_launch_group_scan(ah_context* ctx, const uint32_t* group_total, int64_t ngroups, unsigned long long* group_prefix,
                                 unsigned long long* total, int slot, uint64_t seq);
#endif
