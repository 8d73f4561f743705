// scan_chain.hpp — exclusive scan of up to a few million block totals in ONE launch of chained workgroups.
//
// Used where a kernel has left one total per tile / block / round and the next kernel needs every tile's base plus the grand
// total (string filter: 4096-row tile bytes; string take: 1024-row round bytes; Utf8View cast: out-of-line bytes per block).
// Workgroup b sums its contiguous segment with coalesced loads, publishes FLAG | sum in slots[b], collects the published sums of
// workgroups [0, b) — they were dispatched before it, so the wait cannot deadlock — and writes its segment's bases.  `slots`
// (AH_SCAN_CHAIN_MAX_BLOCKS words) must be zero at launch.  Before round 5 these scans were one workgroup whose threads each
// walked a contiguous run: uncoalesced, 128 dependent steps per thread for the 131 072 tiles of a 2^29-row string filter
// (229 us of its 2.15 ms; 9 us now).
#pragma once
#include "common.hpp"

namespace {

constexpr int AH_SCAN_CHAIN_MAX_BLOCKS = 256;
constexpr unsigned long long AH_SCAN_CHAIN_FLAG = 1ull << 63;

__device__ __forceinline__ unsigned long long block_scan_incl_1024(unsigned long long v, unsigned long long* s_wave, int lane,
                                                                    int wave, unsigned long long* block_total) {
  unsigned long long incl = v;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const unsigned long long u = __shfl_up(incl, k, 64);
    if (lane >= k) incl += u;
  }
  __syncthreads();  // (s_wave of the previous round has been read)
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const unsigned long long x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *block_total = tot;
  return base + incl;
}

// what the last workgroup adds to the read-back beside the grand total (total_out[0]):
//   valid_slots: total_out[1] = the sum of 64 counters (valid rows of the string filter's ranges pass);
//   last_off:    total_out[1] = the offsets buffer's last entry (i32 / i64), the extent of the text a view may read
struct ScanChainExtra {
  const unsigned long long* valid_slots = nullptr;
  const void* last_off = nullptr;
  int last_wide = 0;
};

template <typename T>
__global__ void __launch_bounds__(1024) chained_scan_kernel(const T* totals, int64_t n, int64_t seg, unsigned long long* bases,
                                                            unsigned long long* total_out, ScanChainExtra extra,
                                                            unsigned long long* slots) {
  __shared__ unsigned long long s_wave[16];
  __shared__ unsigned long long s_base;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
  const bool last = b == (int)gridDim.x - 1;
  if (last && wave == 15) {
    if (extra.valid_slots) {
      const unsigned long long v = wave_reduce_add64(extra.valid_slots[lane]);
      if (lane == 0) total_out[1] = v;
    } else if (extra.last_off && lane == 0) {
      total_out[1] = extra.last_wide ? (unsigned long long)*(const long long*)extra.last_off
                                     : (unsigned long long)(long long)*(const int*)extra.last_off;
    }
  }
  const int64_t i0 = (int64_t)b * seg, i1 = i0 + seg < n ? i0 + seg : n;
  // 1. the segment's total
  unsigned long long mine = 0;
  for (int64_t i = i0 + t; i < i1; i += 1024) mine += (unsigned long long)totals[i];
  unsigned long long seg_total = 0;
  (void)block_scan_incl_1024(mine, s_wave, lane, wave, &seg_total);
  if (t == 0) __hip_atomic_store(slots + b, AH_SCAN_CHAIN_FLAG | seg_total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // 2. the totals of the segments in front (wave 0: 64 of them per round)
  if (wave == 0) {
    unsigned long long acc = 0;
    for (int j = lane; j < b; j += 64) {
      unsigned long long x;
      do {
        x = __hip_atomic_load(slots + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (!(x & AH_SCAN_CHAIN_FLAG)) __builtin_amdgcn_s_sleep(1);
      } while (!(x & AH_SCAN_CHAIN_FLAG));
      acc += x & ~AH_SCAN_CHAIN_FLAG;
    }
    acc = wave_reduce_add64(acc);
    if (lane == 0) s_base = acc;
  }
  __syncthreads();
  unsigned long long run = s_base;
  // 3. the segment's bases, 1024 totals per round
  for (int64_t c0 = i0; c0 < i1; c0 += 1024) {
    const int64_t i = c0 + t;
    const unsigned long long v = i < i1 ? (unsigned long long)totals[i] : 0ull;
    unsigned long long tot = 0;
    const unsigned long long incl = block_scan_incl_1024(v, s_wave, lane, wave, &tot);
    if (i < i1) bases[i] = run + incl - v;
    run += tot;
  }
  if (last && t == 0) *total_out = run;
}

// the launch: segments of whole 1024-entry rounds, at most AH_SCAN_CHAIN_MAX_BLOCKS of them
template <typename T>
void ah_launch_chained_scan(ah_context* ctx, const T* totals, int64_t n, unsigned long long* bases, unsigned long long* total_out,
                            ScanChainExtra extra, unsigned long long* slots) {
  const int64_t rounds = std::max<int64_t>(1, ah_ceil_div(n, 1024));
  const int64_t per = ah_ceil_div(rounds, AH_SCAN_CHAIN_MAX_BLOCKS);  // rounds per workgroup
  const int64_t seg = per * 1024;
  const unsigned grid = (unsigned)std::max<int64_t>(1, ah_ceil_div(n, seg));
  chained_scan_kernel<T><<<grid, 1024, 0, ctx->stream>>>(totals, n, seg, bases, total_out, extra, slots);
}

}  // namespace
