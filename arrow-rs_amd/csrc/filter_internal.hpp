// filter_internal.hpp — what filter.hip shares with filter_expr.hip (the lazily evaluated predicate) and coalesce.hip.
#pragma once
#include "common.hpp"

#ifdef __HIPCC__
// FilterPredicate (filter.rs:442-449): predicate bits (borrowed, or owned when built from an expression), count and
// the device-resident prefix tables that replace IterationStrategy::Indices.
struct ah_filter_predicate {
  BitView mask, mask_valid;
  int64_t len = 0;
  int64_t count = 0;
  uint32_t* chunk_prefix = nullptr;
  unsigned long long* group_prefix = nullptr;
  int group_shift = 10;
  void* block = nullptr;  // single pool allocation backing the tables (and the mask of an expression predicate)
  unsigned long long* total_dev = nullptr;  // K on the device (kernels that run before the host has read it)
};

constexpr int AH_FILTER_CHUNK_ROWS = 1024;  // granule of the count pass: 16 mask words

// filter.hip: K2 (scan of the group totals; K lands in pinned slot `slot`, the mailbox is posted with `seq` != 0)
void ah_filter_launch_group_scan(ah_context* ctx, const uint32_t* group_total, int64_t ngroups, unsigned long long* group_prefix,
                                 unsigned long long* total, int slot, uint64_t seq);
#endif
