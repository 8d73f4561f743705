// strings.hip — Utf8 / LargeUtf8 columns through filter and take.
//
// Reference: filter_bytes (arrow-select/src/filter.rs:790-928: offsets rebuilt from the selected
// rows' lengths, bytes copied row by row — null slots with a non-zero length are copied too) and
// take_bytes (arrow-select/src/take.rs:499-627: output nulls get zero-length slots; i32 offsets
// past i32::MAX => ArrowError::OffsetOverflowError(capacity)).
//
// MI355X design: both are "ranges -> scan of block totals -> gather", three launches and one wait before the gather
// (the byte total sizes the data buffer):
//   filter: string_filter_ranges_kernel / chained_scan_kernel / string_filter_gather_kernel (F1-F3 below);
//   take:   T1 take_ranges_kernel: index -> source start (zero length under an output null), the output validity word, and —
//              a workgroup round being 1024 consecutive output rows — the rows' byte offsets INSIDE the round plus the
//              round's byte total (rounds 1-4 wrote [start, end) pairs and ran a three-kernel scan over them: 0.47 ms of
//              the 4.1 ms step at 5.4e7 indices);
//           T2 chained_scan_kernel (scan_chain.hpp) over the round totals -> round bases, grand total;
//           T3 take_gather_rows_kernel: final offset = round base + local offset (written out), one output row per thread,
//              unaligned 8-byte chunks with an overlapping tail (rows <= 64 B); longer rows are copied cooperatively by
//              the workgroup.  (The first version walked output bytes with a binary search per byte: 0.32 ms per 120 MB.)
#include "common.hpp"
#include "filter_internal.hpp"
#include "scan_chain.hpp"

#include <type_traits>

#ifndef AH_SFR_CAP
#define AH_SFR_CAP 1024  // selected rows staged per round of the ranges kernel: 18 KiB of LDS, 7 workgroups per CU (2048: 36 KiB, 4 per CU, 0.343 ms against 0.309 per 2^27 rows)
#endif

namespace {

// ---- filter_bytes (filter.rs:790-928) as three launches: ranges -> tile-byte scan -> gather
//
// F1 string_filter_ranges_kernel: ONE pass over the offsets (it used to be two runs of the primitive scatter kernel,
//    over offsets[0..n) and offsets[1..n+1), then a three-kernel scan over the K selected rows).  A tile is 4096 rows;
//    every thread owns 16 consecutive rows = a quarter of one predicate word and loads its 17 offsets with 16-byte loads.
//    A block scan over (selected rows, selected bytes) per thread gives every selected row its rank in the tile and its
//    byte offset INSIDE the tile's output; (start, local offset) pairs are compacted through a 16 KiB LDS stage (1024 rows per round) and leave
//    as coalesced stores at the tile's first output row (known from the predicate's prefix tables).  The tile's byte
//    total goes to tile_bytes[tile].
// F2 chained_scan_kernel (scan_chain.hpp): exclusive scan of the <= n / 4096 tile totals (one launch, <= 256 chained workgroups) -> tile_base, grand total
//    (the one number the host waits for: it sizes the data buffer).
// F3 string_filter_gather_kernel: one workgroup per tile again: new offset = tile_base + local offset, bytes copied row by
//    row (unaligned 8-byte chunks with an overlapping tail; rows > 64 B cooperatively).
//    With HAS_VALID the same pass compacts the column's validity (filter_nulls, filter.rs:512-532): the selected rows'
//    validity flags ride through the stage, are re-packed with __ballot aligned to the output's 64-bit words and merged
//    with atomicOr (pre-zeroed bitmap), the valid rows are counted into *valid_count (it was a separate bit-only scatter
//    over the predicate, 107 us at 2^27 rows, plus a popcount launch and a host wait).
// PT = the pairs' element type.  OFF: (absolute start, local offset).  uint32_t with 64-bit offsets ("narrow pairs", round 5):
// (start - the tile's first offset, local offset) — half the bytes of the pairs' round trip through HBM (0.86 of ~10.7 GB per
// 2^29 rows at 10 % selected); a tile whose 4096 rows span >= 4 GiB of text cannot be encoded and raises *wide_needed (the host
// then reruns the call with PT = OFF).
template <typename OFF, typename PT, bool VEC, bool HAS_VALID>
__global__ void __launch_bounds__(256) string_filter_ranges_kernel(const OFF* offsets, BitView mask, BitView mask_valid,
                                                                   int64_t len, const uint32_t* chunk_prefix,
                                                                   const unsigned long long* group_prefix, int group_shift,
                                                                   PT* starts, PT* loffs, unsigned long long* tile_bytes,
                                                                   BitView vvalid, unsigned long long* out_valid,
                                                                   unsigned long long* valid_count,
                                                                   unsigned long long* wide_needed) {
  constexpr int T = 4096, NW = 64, R = 16, CAP = AH_SFR_CAP;
  constexpr bool NARROW = sizeof(PT) < sizeof(OFF);
  __shared__ OFF s_tile_first;
  __shared__ uint64_t s_m[NW];
  __shared__ uint64_t s_v[HAS_VALID ? NW : 1];
  __shared__ uint32_t s_base[NW];
  __shared__ uint32_t s_total;
  __shared__ unsigned long long s_wbytes[4];
  __shared__ PT s_start[CAP];
  __shared__ PT s_loff[CAP];
  __shared__ uint8_t s_flag[HAS_VALID ? CAP : 1];
  __shared__ uint32_t s_vc[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * T;
  const int64_t r0 = row0 + (int64_t)t * R;
  // the thread's offsets go out first
  OFF o[R + 1];
  OFF olast;  // the thread's last offset inside the array
  if (r0 + R <= len) {
    if constexpr (VEC) {
      constexpr int PV = 16 / sizeof(OFF);  // offsets per 16-byte load
      struct alignas(16) V16 { OFF e[PV]; };
#pragma unroll
      for (int k = 0; k < R / PV; ++k) {
        const V16 v = *(const V16*)(offsets + r0 + k * PV);
#pragma unroll
        for (int e = 0; e < PV; ++e) o[k * PV + e] = v.e[e];
      }
      o[R] = offsets[r0 + R];
    } else {
#pragma unroll
      for (int e = 0; e <= R; ++e) o[e] = offsets[r0 + e];
    }
    olast = o[R];
  } else {
    olast = 0;
#pragma unroll
    for (int e = 0; e <= R; ++e) {
      o[e] = r0 + e <= len ? offsets[r0 + e] : (OFF)0;
      if (r0 + e <= len) olast = o[e];
    }
  }
  if (NARROW && t == 0) s_tile_first = o[0];
  if (wave == 0) {
    uint64_t m = 0, vv = 0;
    const int64_t s = row0 + ((int64_t)lane << 6);
    if (s < len) {
      m = bv_fetch64(mask, s, len);
      if (mask_valid.words) m &= bv_fetch64(mask_valid, s, len);
      if constexpr (HAS_VALID) vv = bv_fetch64(vvalid, s, len);
    }
    const int c = __popcll(m);
    const int incl = wave_scan_incl(c);
    s_m[lane] = m;
    if constexpr (HAS_VALID) s_v[lane] = vv;
    s_base[lane] = (uint32_t)(incl - c);
    if (lane == 63) s_total = (uint32_t)incl;
  }
  __syncthreads();
  const int sh = (t * R) & 63;
  const uint64_t word = s_m[(t * R) >> 6];
  const uint32_t bits = (uint32_t)(word >> sh) & 0xFFFFu;
  uint32_t vbits = 0;
  if constexpr (HAS_VALID) vbits = (uint32_t)(s_v[(t * R) >> 6] >> sh) & 0xFFFFu;
  // bytes this thread's selected rows contribute, and their exclusive prefix over the workgroup (thread order = row order)
  unsigned long long mine = 0;
#pragma unroll
  for (int e = 0; e < R; ++e)
    if ((bits >> e) & 1u) mine += (unsigned long long)(o[e + 1] - o[e]);
  OFF tile_first = 0;
  if constexpr (NARROW) {
    tile_first = s_tile_first;
    // a thread with selected rows whose LAST offset lies 4 GiB or more behind the tile's first one: a start or a local offset
    // may not fit (local offset <= start - tile_first: the selected bytes in front of a row are part of the span in front of
    // it).  Conservative — the thread's last row need not be a selected one — and one compare per thread.
    if (bits && ((unsigned long long)(olast - tile_first) >> 32)) atomicOr(wide_needed, 1ull);
    // from here on the thread's offsets are relative to the tile's first one, in place (lengths are differences: unchanged)
#pragma unroll
    for (int e = 0; e <= R; ++e) o[e] -= tile_first;
  }
  unsigned long long incl = mine;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const unsigned long long u = __shfl_up(incl, k, 64);
    if (lane >= k) incl += u;
  }
  if (lane == 63) s_wbytes[wave] = incl;
  __syncthreads();
  unsigned long long byte_off = incl - mine;
  for (int w = 0; w < wave; ++w) byte_off += s_wbytes[w];
  const int total = (int)s_total;
  if (t == 0) tile_bytes[blockIdx.x] = s_wbytes[0] + s_wbytes[1] + s_wbytes[2] + s_wbytes[3];
  if (total == 0) return;
  const int64_t chunk0 = row0 / AH_FILTER_CHUNK_ROWS;
  const int64_t P = (int64_t)group_prefix[chunk0 >> group_shift] + chunk_prefix[chunk0];  // the tile's first output row
  const int rank0 = (int)(s_base[(t * R) >> 6] + __popcll(word & ((1ull << sh) - 1ull)));
  int vc = 0;
  for (int p0 = 0; p0 < total; p0 += CAP) {  // tiles with more than CAP selected rows take a second round
    const int cnt = (total - p0) < CAP ? (total - p0) : CAP;
    if (p0) __syncthreads();
    int pos = rank0 - p0;
    unsigned long long bo = byte_off;
#pragma unroll
    for (int e = 0; e < R; ++e) {
      if ((bits >> e) & 1u) {
        if ((unsigned)pos < (unsigned)cnt) {
          s_start[pos] = (PT)o[e];  // (narrow pairs: o[] is tile-relative by now)
          s_loff[pos] = (PT)bo;
          if constexpr (HAS_VALID) s_flag[pos] = (uint8_t)((vbits >> e) & 1u);
        }
        bo += (unsigned long long)(o[e + 1] - o[e]);
        ++pos;
      }
    }
    __syncthreads();
    for (int q = t; q < cnt; q += 256) {
      starts[P + p0 + q] = s_start[q];
      loffs[P + p0 + q] = s_loff[q];
    }
    if constexpr (HAS_VALID) {  // validity words of output rows [P + p0, + cnt): ballots aligned to the global words
      const int64_t cb = P + p0, g0 = cb & ~63ll;
      const int lead = (int)(cb - g0), span64 = (lead + cnt + 63) & ~63;
      for (int q = t; q < span64; q += 256) {
        const int j = q - lead;
        const int f = (j >= 0 && j < cnt) ? (int)s_flag[j] : 0;
        const uint64_t w = __ballot(f);
        if (lane == 0) {
          if (w) atomicOr(&out_valid[(g0 + q) >> 6], (unsigned long long)w);
          vc += __popcll(w);
        }
      }
    }
  }
  if constexpr (HAS_VALID) {
    if (lane == 0) s_vc[wave] = (uint32_t)vc;
    __syncthreads();
    // 64 counters, one add per tile (32 Ki same-address atomics cost 0.2 ms: ~12 ns each, serialized); the tile scan folds them
    if (t == 0) ah_count_add(valid_count, (unsigned long long)(s_vc[0] + s_vc[1] + s_vc[2] + s_vc[3]));
  }
}

// F1 for sparse selections (K * 32 <= len; the reference bench's "filter context string low selectivity (kept 1/1024)"):
// the tiled kernel above reads ALL offsets — 1 GB per 2^27 rows, 0.31 ms at any selectivity — and is a per-tile chain of
// four barriers.  Here a 4096-row tile belongs to ONE WAVE (lane l = predicate word l), only the offsets of selected rows
// are loaded, the (start, local offset) pairs go straight to their output rows, validity bits are collected in a
// wave-private LDS strip and leave as one atomicOr per output word; valid rows are counted from the output bitmap
// afterwards (bitmap_count_to_slots_kernel) — per-tile counter atomics serialise across the XCDs.
template <typename OFF, bool HAS_VALID>
__global__ void __launch_bounds__(256) string_filter_ranges_sparse_kernel(const OFF* offsets, BitView mask, BitView mask_valid, int64_t len,
                                                                          const uint32_t* chunk_prefix,
                                                                          const unsigned long long* group_prefix, int group_shift,
                                                                          OFF* starts, OFF* loffs, unsigned long long* tile_bytes,
                                                                          BitView vvalid, unsigned long long* out_valid,
                                                                          int64_t ntiles) {
  __shared__ unsigned long long s_w[HAS_VALID ? 4 : 1][HAS_VALID ? 66 : 1];
  const int lane = threadIdx.x & 63, wave = ah_uniform((int)(threadIdx.x >> 6));
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int64_t row0 = tile * 4096, s = row0 + ((int64_t)lane << 6);
  uint64_t m = 0, v = 0;
  if (s < len) {
    m = bv_fetch64(mask, s, len);
    if (mask_valid.words) m &= bv_fetch64(mask_valid, s, len);
    if constexpr (HAS_VALID) v = bv_fetch64(vvalid, s, len);
  }
  const int c = __popcll(m);
  const int incl = wave_scan_incl(c);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  if (total == 0) {
    if (lane == 0) tile_bytes[tile] = 0;
    return;
  }
  const OFF* op = offsets + s;
  unsigned long long mine = 0;  // bytes of this lane's selected rows
  for (uint64_t mm = m; mm; mm &= mm - 1) {
    const int b = __builtin_ctzll(mm);
    mine += (unsigned long long)(op[b + 1] - op[b]);
  }
  unsigned long long bincl = mine;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const unsigned long long u = __shfl_up(bincl, k, 64);
    if (lane >= k) bincl += u;
  }
  const unsigned long long tbytes = __shfl(bincl, 63, 64);
  if (lane == 0) tile_bytes[tile] = tbytes;
  const int64_t chunk0 = row0 / AH_FILTER_CHUNK_ROWS;
  const int64_t P = (int64_t)group_prefix[chunk0 >> group_shift] + chunk_prefix[chunk0];  // the tile's first output row
  const int64_t wb = P >> 6;
  if constexpr (HAS_VALID) {
    s_w[wave][lane] = 0;
    if (lane < 2) s_w[wave][64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  int64_t pos = P + (incl - c);
  unsigned long long bo = bincl - mine;  // byte offset of this lane's first selected row inside the tile's output
  for (uint64_t mm = m; mm; mm &= mm - 1) {
    const int b = __builtin_ctzll(mm);
    const OFF o0 = op[b], o1 = op[b + 1];
    starts[pos] = o0;
    loffs[pos] = (OFF)bo;
    bo += (unsigned long long)(o1 - o0);
    if constexpr (HAS_VALID) {
      if ((v >> b) & 1ull) atomicOr(&s_w[wave][(int)((pos >> 6) - wb)], 1ull << (pos & 63));
    }
    ++pos;
  }
  if constexpr (HAS_VALID) {
    __builtin_amdgcn_wave_barrier();
    const int nw = (int)(((P + total - 1) >> 6) - wb) + 1;  // <= 65
    for (int j = lane; j < nw; j += 64) {
      const unsigned long long w = s_w[wave][j];
      if (w) atomicOr(&out_valid[wb + j], w);
    }
  }
}

// set bits among the first `nbits` bits of `bits`, added block by block to slots[0..64)
__global__ void __launch_bounds__(256) bitmap_count_to_slots_kernel(const unsigned long long* bits, int64_t nbits,
                                                                    unsigned long long* slots) {
  const int64_t nwords = (nbits + 63) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
    unsigned long long x = bits[w];
    if (w == nwords - 1 && (nbits & 63)) x &= (1ull << (nbits & 63)) - 1ull;
    acc += (unsigned long long)__popcll(x);
  }
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s_acc[4];
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3];
    if (t) atomicAdd(&slots[blockIdx.x & 63], t);
  }
}

// F2 / T2: exclusive scan of the tile (round) byte totals, grand total -> total[0]; with `valid_slots`, total[1] = valid rows
static void launch_string_tile_scan(ah_context* ctx, const unsigned long long* tile_bytes, int64_t ntiles,
                                    unsigned long long* tile_base, unsigned long long* total,
                                    const unsigned long long* valid_slots, unsigned long long* slots) {
  ScanChainExtra extra;
  extra.valid_slots = valid_slots;
  ah_launch_chained_scan<unsigned long long>(ctx, tile_bytes, ntiles, tile_base, total, extra, slots);
}

__device__ __forceinline__ void copy8(uint8_t* d, const uint8_t* s) {  // unaligned 8-byte move
  unsigned long long v;
  __builtin_memcpy(&v, s, 8);
  __builtin_memcpy(d, &v, 8);
}
__device__ __forceinline__ void copy4(uint8_t* d, const uint8_t* s) {
  unsigned int v;
  __builtin_memcpy(&v, s, 4);
  __builtin_memcpy(d, &v, 4);
}

// T3.  One output row per thread: rows up to 64 bytes are moved with unaligned 8-byte chunks (the last
// chunk overlaps backwards instead of a byte tail — gfx950 global accesses need no alignment);
// consecutive lanes write consecutive regions, so a wave's stores still cover one contiguous span.
// Longer rows are queued in LDS and copied by the whole workgroup, 8 bytes per thread per step.
// Row j's bytes go to round_base[j / 1024] + loc[j]; that sum is also the row's new offset.
template <typename OFF>
__global__ void __launch_bounds__(256) take_gather_rows_kernel(const uint8_t* src, const OFF* starts, const OFF* loc,
                                                               const unsigned long long* round_base,
                                                               const unsigned long long* total, int64_t k, OFF* dst_off,
                                                               uint8_t* dst) {
  __shared__ unsigned long long s_rs[256], s_rd[256], s_rl[256];
  __shared__ int s_nlong;
  const int t = threadIdx.x;
  const int64_t j = (int64_t)blockIdx.x * 256 + t;
  if (t == 0) s_nlong = 0;
  __syncthreads();
  if (j < k) {
    const int64_t r = j >> 10;
    const unsigned long long rb = round_base[r];
    const unsigned long long s0 = (unsigned long long)starts[j];
    // (a local offset is kept in the offset type: it wraps for i32 offsets only when the total overflows, which is an error
    //  before this kernel runs)
    const unsigned long long d0 = rb + (unsigned long long)(typename std::make_unsigned<OFF>::type)loc[j];
    unsigned long long d1;
    if (j + 1 == k) d1 = *total;
    else if (((j + 1) & 1023) == 0) d1 = round_base[r + 1];
    else d1 = rb + (unsigned long long)(typename std::make_unsigned<OFF>::type)loc[j + 1];
    const unsigned long long len = d1 - d0;
    dst_off[j] = (OFF)d0;
    if (j + 1 == k) dst_off[k] = (OFF)d1;
    const uint8_t* sp = src + s0;
    uint8_t* dp = dst + d0;
    if (len > 64) {
      const int q = atomicAdd(&s_nlong, 1);
      s_rs[q] = s0, s_rd[q] = d0, s_rl[q] = len;
    } else if (len >= 8) {
      for (unsigned o = 0; o + 8 <= (unsigned)len; o += 8) copy8(dp + o, sp + o);
      if (len & 7) copy8(dp + len - 8, sp + len - 8);
    } else if (len >= 4) {
      copy4(dp, sp);
      copy4(dp + len - 4, sp + len - 4);
    } else {
      for (unsigned o = 0; o < (unsigned)len; ++o) dp[o] = sp[o];
    }
  }
  __syncthreads();
  const int nlong = s_nlong;
  for (int q = 0; q < nlong; ++q) {
    const unsigned long long rs = s_rs[q], rd = s_rd[q], rl = s_rl[q];
    const unsigned long long whole = rl & ~7ull;
    for (unsigned long long o = (unsigned long long)t * 8; o < whole; o += 2048) copy8(dst + rd + o, src + rs + o);
    if ((unsigned long long)t < rl - whole) dst[rd + whole + t] = src[rs + whole + t];
  }
}

// F3: the tile's selected rows [P, P + C): new offsets and bytes.  Same copy scheme as gather_bytes_kernel.
// With narrow pairs (PT = uint32_t under 64-bit offsets) a start is relative to the tile's first offset, read here once.
template <typename OFF, typename PT>
__global__ void __launch_bounds__(256) string_filter_gather_kernel(const uint8_t* src, const OFF* offsets, const PT* starts,
                                                                   const PT* loffs, const unsigned long long* tile_bytes,
                                                                   const unsigned long long* tile_base, int64_t len, int64_t K,
                                                                   const uint32_t* chunk_prefix, const unsigned long long* group_prefix,
                                                                   int group_shift, OFF* dst_off, uint8_t* dst) {
  __shared__ int s_long[256];
  __shared__ int s_nlong;
  const int t = threadIdx.x;
  const int64_t ntiles = (len + 4095) / 4096, tile = blockIdx.x;
  using UPT = typename std::make_unsigned<PT>::type;
  const unsigned long long tile_first = sizeof(PT) < sizeof(OFF) ? (unsigned long long)offsets[tile * 4096] : 0ull;
  auto first_row = [&](int64_t tl) -> int64_t {
    if (tl >= ntiles) return K;
    const int64_t c0 = tl * 4096 / AH_FILTER_CHUNK_ROWS;
    return (int64_t)group_prefix[c0 >> group_shift] + chunk_prefix[c0];
  };
  const int64_t P = first_row(tile);
  const int C = (int)(first_row(tile + 1) - P);
  const unsigned long long base = tile_base[tile], tbytes = tile_bytes[tile];
  if (tile == ntiles - 1 && t == 0) dst_off[K] = (OFF)(base + tbytes);
  for (int j0 = 0; j0 < C; j0 += 256) {
    const int j = j0 + t;
    if (t == 0) s_nlong = 0;
    __syncthreads();
    if (j < C) {
      const unsigned long long s0 = tile_first + (unsigned long long)(UPT)starts[P + j], lo = (unsigned long long)(UPT)loffs[P + j];
      const unsigned long long hi = j + 1 < C ? (unsigned long long)(UPT)loffs[P + j + 1] : tbytes;
      const unsigned long long n = hi - lo, d0 = base + lo;
      dst_off[P + j] = (OFF)d0;
      const uint8_t* sp = src + s0;
      uint8_t* dp = dst + d0;
      if (n > 64) {
        s_long[atomicAdd(&s_nlong, 1)] = j;
      } else if (n >= 8) {
        for (unsigned o = 0; o + 8 <= (unsigned)n; o += 8) copy8(dp + o, sp + o);
        if (n & 7) copy8(dp + n - 8, sp + n - 8);
      } else if (n >= 4) {
        copy4(dp, sp);
        copy4(dp + n - 4, sp + n - 4);
      } else {
        for (unsigned o = 0; o < (unsigned)n; ++o) dp[o] = sp[o];
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int q = 0; q < nlong; ++q) {
      const int r = s_long[q];
      const unsigned long long rs = tile_first + (unsigned long long)(UPT)starts[P + r], lo = (unsigned long long)(UPT)loffs[P + r];
      const unsigned long long hi = r + 1 < C ? (unsigned long long)(UPT)loffs[P + r + 1] : tbytes;
      const unsigned long long rl = hi - lo, rd = base + lo, whole = rl & ~7ull;
      for (unsigned long long o = (unsigned long long)t * 8; o < whole; o += 2048) copy8(dst + rd + o, src + rs + o);
      if ((unsigned long long)t < rl - whole) dst[rd + whole + t] = src[rs + whole + t];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ unsigned long long wave_scan_incl64(unsigned long long v, int lane) {
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const unsigned long long u = __shfl_up(v, k, 64);
    if (lane >= k) v += u;
  }
  return v;
}

// T1.  indices -> source starts (take_bytes, take.rs:499-627) and, in the same pass, the output validity
// (take_nulls, take.rs:418-430: index validity AND values.validity[index]) as one ballot word per wave.
// Output nulls produce empty rows (take.rs:553-577); a valid out-of-bounds index is reported through
// first_oob.  counters[0] = first OOB position, counters[1] = number of valid output slots.
// A workgroup round is 1024 consecutive output rows (wave w: rows [256 w, 256 w + 256), four gathers in flight per lane):
// the rows' lengths are scanned inside the round — loc[i] = bytes of the round's rows in front of row i — and
// round_total[round] gets the round's bytes, so what is left of the offsets scan is one pass over n / 1024 totals.
template <typename OFF, typename IDX>
__global__ void __launch_bounds__(256) take_ranges_kernel(const OFF* offsets, int64_t nvalues, const IDX* idx,
                                                          int64_t n, BitView ivalid, BitView vvalid,
                                                          unsigned long long* out_valid, OFF* starts, OFF* loc,
                                                          unsigned long long* round_total,
                                                          unsigned long long* counters) {
  constexpr int KU = 4;  // independent gathers in flight per lane: the offsets pair and the validity bit of 4 rows
  __shared__ unsigned long long s_wtot[2][4];
  unsigned long long oob = ~0ull, nvalid = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int par = 0;
  for (int64_t base = (int64_t)blockIdx.x * (256 * KU); base < n; base += (int64_t)gridDim.x * (256 * KU), par ^= 1) {
    const int64_t wbase = base + wave * (64 * KU);
    uint64_t ix[KU];
    bool live[KU], inb[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int64_t i = wbase + k * 64 + lane;
      live[k] = i < n && bv_get(ivalid, i);
      ix[k] = 0;
      if (live[k]) {
        IDX raw = idx[i];
        if constexpr (sizeof(IDX) <= 2 && std::is_signed<IDX>::value) ix[k] = (uint32_t)(int32_t)raw;
        else if constexpr (sizeof(IDX) == 4) ix[k] = (uint32_t)raw;
        else ix[k] = (uint64_t)raw;
      }
      inb[k] = ix[k] < (uint64_t)nvalues;
    }
    OFF s[KU], e[KU];
    int vb[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {  // the three gathers of a row do not depend on each other
      s[k] = 0;
      e[k] = 0;
      vb[k] = 0;
      if (live[k] && inb[k]) {
        s[k] = offsets[ix[k]];
        e[k] = offsets[ix[k] + 1];
        vb[k] = bv_get(vvalid, (int64_t)ix[k]);
      }
    }
    unsigned long long pre[KU], wrun = 0;  // bytes of this wave's rows in front of row k * 64 + lane; the wave's bytes
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int64_t i = wbase + k * 64 + lane;
      if (live[k] && !inb[k] && (unsigned long long)i < oob) oob = (unsigned long long)i;
      const unsigned long long len = vb[k] ? (unsigned long long)(e[k] - s[k]) : 0ull;
      const unsigned long long incl = wave_scan_incl64(len, lane);
      pre[k] = wrun + incl - len;
      wrun += __shfl(incl, 63, 64);
      const unsigned long long w = __ballot(vb[k]);
      if (lane == 0 && wbase + k * 64 < n) {
        if (out_valid) out_valid[(wbase + k * 64) >> 6] = w;
        nvalid += __popcll(w);
      }
    }
    if (lane == 0) s_wtot[par][wave] = wrun;
    __syncthreads();  // (every thread of the workgroup runs the same rounds; s_wtot alternates, so one barrier per round)
    unsigned long long wfront = 0, rtot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned long long x = s_wtot[par][w];
      if (w < wave) wfront += x;
      rtot += x;
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int64_t i = wbase + k * 64 + lane;
      if (i < n) {
        starts[i] = vb[k] ? s[k] : (OFF)0;
        loc[i] = (OFF)(wfront + pre[k]);
      }
    }
    if (threadIdx.x == 0) round_total[base >> 10] = rtot;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(oob, o, 64);
    oob = other < oob ? other : oob;
  }
  if (lane == 0) {
    if (oob != ~0ull) atomicMin(&counters[0], oob);
    if (nvalid) atomicAdd(&counters[1], nvalid);
  }
}

template <typename OFF>
ah_status launch_take_ranges(ah_context* ctx, const ah_array_view* values, const ah_array_view* indices,
                             BitView ivalid, BitView vvalid, unsigned long long* out_valid, OFF* starts, OFF* loc,
                             unsigned long long* round_total, unsigned long long* counters) {
  const int64_t n = indices->length;
  int g = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(n, 1024), 8192));
  const OFF* off = (const OFF*)values->offsets;
#define AH_TR(IDX) take_ranges_kernel<OFF, IDX><<<g, 256, 0, ctx->stream>>>(off, values->length, (const IDX*)indices->values, n, ivalid, vvalid, out_valid, starts, loc, round_total, counters)
  switch (indices->type) {
    case AH_INT8: AH_TR(int8_t); break;
    case AH_UINT8: AH_TR(uint8_t); break;
    case AH_INT16: AH_TR(int16_t); break;
    case AH_UINT16: AH_TR(uint16_t); break;
    case AH_INT32: AH_TR(int32_t); break;
    case AH_UINT32: AH_TR(uint32_t); break;
    case AH_INT64: AH_TR(int64_t); break;
    case AH_UINT64: AH_TR(uint64_t); break;
    default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "Take only supported for integers, got %s", ah_type_name(indices->type));
  }
#undef AH_TR
  return AH_OK;
}

}  // namespace

// filter_bytes (filter.rs:790-928) + filter_nulls (:512-532) for the rows `p` selects: offsets, data and — when `vvalid` has
// words — the compacted validity of the result (dropped again when it has no nulls, :523-525).
template <typename OFF, typename PT>
static ah_status filter_bytes_t(ah_context* ctx, const ah_filter_predicate* p, const ah_array_view* values, BitView vvalid,
                                bool sparse, ah_array_out* out, bool* wide_needed) {
  constexpr bool NARROW = sizeof(PT) < sizeof(OFF);
  const int64_t K = p->count, len = p->len, ntiles = ah_ceil_div(len, 4096);
  const OFF* offsets = (const OFF*)values->offsets;
  const bool hv = vvalid.words != nullptr;
  const size_t ob = (size_t)(K + 1) * sizeof(OFF), kb = (((size_t)K * sizeof(PT)) + 15) & ~(size_t)15, nbytes = hv ? ah_bitmap_bytes(K) : 0;
  // starts | local offsets | tile bytes | tile bases | {total bytes, valid rows} | {pairs too narrow, -} | 64 valid-row counters | tile-scan slots
  char* tmp = nullptr;
  AH_TRY(ah_pool_alloc(ctx, 2 * kb + (size_t)(2 * ntiles + 4 + 64 + AH_SCAN_CHAIN_MAX_BLOCKS) * 8, (void**)&tmp));
  void* nb = nullptr;
  if (hv) {
    const ah_status as = ah_out_alloc(ctx, nbytes, &nb);
    if (as != AH_OK) {
      ah_pool_free(ctx, tmp);
      return as;
    }
    hipMemsetAsync(nb, 0, nbytes, ctx->stream);
  }
  PT* starts = (PT*)tmp;
  PT* loffs = (PT*)(tmp + kb);
  unsigned long long* tile_bytes = (unsigned long long*)(tmp + 2 * kb);
  unsigned long long* tile_base = tile_bytes + ntiles;
  unsigned long long* total = tile_base + ntiles;  // [0] byte total, [1] valid rows (both written by the tile scan), [2] pairs too narrow
  unsigned long long* vslots = total + 4;
  unsigned long long* scan_slots = vslots + 64;
  hipMemsetAsync(total + 2, 0, (size_t)(2 + 64 + AH_SCAN_CHAIN_MAX_BLOCKS) * 8, ctx->stream);
  const bool vec = (((uintptr_t)offsets) & 15) == 0;
  {  // ranges + tile scan: one profiled span
  ah_prof_scope ps(ctx, "string_filter_ranges");
  if (sparse) {
    if constexpr (!NARROW) {
      const unsigned grid = (unsigned)ah_ceil_div(ntiles, 4);
      if (hv) {
        string_filter_ranges_sparse_kernel<OFF, true><<<grid, 256, 0, ctx->stream>>>(offsets, p->mask, p->mask_valid, len, p->chunk_prefix,
                                                                                    p->group_prefix, p->group_shift, starts, loffs,
                                                                                    tile_bytes, vvalid, (unsigned long long*)nb, ntiles);
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ah_ceil_div((K + 63) >> 6, 256 * 4)));
        bitmap_count_to_slots_kernel<<<gx, 256, 0, ctx->stream>>>((const unsigned long long*)nb, K, vslots);
      } else {
        string_filter_ranges_sparse_kernel<OFF, false><<<grid, 256, 0, ctx->stream>>>(offsets, p->mask, p->mask_valid, len, p->chunk_prefix,
                                                                                     p->group_prefix, p->group_shift, starts, loffs,
                                                                                     tile_bytes, vvalid, nullptr, ntiles);
      }
    }
  } else {
#define AH_SFR(VEC, HV)                                                                                                          \
  string_filter_ranges_kernel<OFF, PT, VEC, HV><<<(unsigned)ntiles, 256, 0, ctx->stream>>>(                                      \
      offsets, p->mask, p->mask_valid, len, p->chunk_prefix, p->group_prefix, p->group_shift, starts, loffs, tile_bytes, vvalid, \
      (unsigned long long*)nb, vslots, total + 2)
    if (vec && hv) AH_SFR(true, true);
    else if (vec) AH_SFR(true, false);
    else if (hv) AH_SFR(false, true);
    else AH_SFR(false, false);
#undef AH_SFR
  }
  launch_string_tile_scan(ctx, tile_bytes, ntiles, tile_base, total, hv ? vslots : nullptr, scan_slots);
  }
  hipError_t e = ah_d2h_wait(ctx, ctx->pinned, total, 24);  // the one wait before the gather: byte total, valid rows, pair overflow
  if (e != hipSuccess) {
    ah_pool_free(ctx, tmp);
    ah_out_free(ctx, nb, nbytes);
    return ah_fail(ctx, AH_HIP_ERROR, "string filter ranges failed: %s", hipGetErrorString(e));
  }
  const uint64_t total_bytes = ctx->pinned[0];
  const int64_t nulls = hv ? K - (int64_t)ctx->pinned[1] : 0;
  if (NARROW && ctx->pinned[2]) {  // a tile spans >= 4 GiB of text: the caller reruns with full-width pairs
    ah_pool_free(ctx, tmp);
    ah_out_free(ctx, nb, nbytes);
    *wide_needed = true;
    return AH_OK;
  }
  if (sizeof(OFF) == 4 && total_bytes > (uint64_t)INT32_MAX) {
    ah_pool_free(ctx, tmp);
    ah_out_free(ctx, nb, nbytes);
    return ah_fail(ctx, AH_PANIC, "illegal offset range");  // filter.rs:838
  }
  void *offs = nullptr, *data = nullptr;
  ah_status st = ah_out_alloc(ctx, ob, &offs);
  if (st == AH_OK) st = ah_out_alloc(ctx, (size_t)total_bytes, &data);
  if (st != AH_OK) {
    ah_out_free(ctx, offs, ob);
    ah_out_free(ctx, nb, nbytes);
    ah_pool_free(ctx, tmp);
    return st;
  }
  {
    ah_prof_scope ps(ctx, "string_gather_bytes");
    string_filter_gather_kernel<OFF, PT><<<(unsigned)ntiles, 256, 0, ctx->stream>>>((const uint8_t*)values->values, offsets, starts, loffs,
                                                                                    tile_bytes, tile_base, len, K, p->chunk_prefix,
                                                                                    p->group_prefix, p->group_shift, (OFF*)offs, (uint8_t*)data);
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  ah_pool_free(ctx, tmp);
  if (e != hipSuccess) {
    ah_out_free(ctx, offs, ob);
    ah_out_free(ctx, data, (size_t)total_bytes);
    ah_out_free(ctx, nb, nbytes);
    return ah_fail(ctx, AH_HIP_ERROR, "string gather failed: %s", hipGetErrorString(e));
  }
  out->offsets = offs;
  out->offsets_bytes = (int64_t)ob;
  out->values = data;
  out->values_bytes = (int64_t)total_bytes;
  if (hv && nulls > 0) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)nbytes;
    out->null_count = nulls;
  } else {
    ah_out_free(ctx, nb, nbytes);  // filter_nulls :523-525 -> None
  }
  return AH_OK;
}
ah_status ah_string_filter_bytes(ah_context* ctx, const ah_filter_predicate* p, const ah_array_view* values, BitView vvalid,
                                 ah_array_out* out) {
  bool sparse = p->count * 32 <= p->len;  // as for the primitive scatter (filter.hip: use_sparse); AH_FILTER_SPARSE=0 / 1 forces
  if (const char* env = getenv("AH_FILTER_SPARSE")) {
    if (env[0] == '0') sparse = false;
    if (env[0] == '1') sparse = true;
  }
  bool wide = false;
  if (values->type != AH_LARGE_UTF8) return filter_bytes_t<int32_t, int32_t>(ctx, p, values, vvalid, sparse, out, &wide);
  // 64-bit offsets: 4-byte (tile-relative start, local offset) pairs unless a tile spans >= 4 GiB (AH_STRING_PAIRS=wide: A/B runs)
  const char* pe = getenv("AH_STRING_PAIRS");
  if (!sparse && !(pe && pe[0] == 'w')) {
    AH_TRY((filter_bytes_t<int64_t, uint32_t>(ctx, p, values, vvalid, false, out, &wide)));
    if (!wide) return AH_OK;
  }
  return filter_bytes_t<int64_t, int64_t>(ctx, p, values, vvalid, sparse, out, &wide);
}

// take_bytes (arrow-select/src/take.rs:499-627)
ah_status ah_take_bytes(ah_context* ctx, const ah_array_view* values, const ah_array_view* indices,
                        ah_array_out* out) {
  const bool large = values->type == AH_LARGE_UTF8;
  const int64_t n = indices->length;
  const size_t ow = large ? 8 : 4;
  out->type = values->type;
  if (n == 0) {  // take_impl :215-217 -> new_empty_array: offsets = [0]
    void* offs = nullptr;
    AH_TRY(ah_out_alloc(ctx, ow, &offs));
    hipMemsetAsync(offs, 0, ow, ctx->stream);
    AH_HIP(ctx, ah_stream_wait(ctx));
    out->offsets = offs;
    out->offsets_bytes = (int64_t)ow;
    return AH_OK;
  }
  if (!values->offsets) return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
  // take_nulls (take.rs:418-430) is fused into the ranges pass: one kernel gathers the offsets pair and the
  // validity bit of each index, one 16-byte read-back brings the first OOB position and the valid count
  int64_t val_nulls = 0, idx_nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, values, &val_nulls));
  AH_TRY(ah_resolve_null_count(ctx, indices, &idx_nulls));
  const bool values_nullable = values->validity && val_nulls > 0;
  const BitView none{nullptr, 0};
  const BitView ivalid = (indices->validity && idx_nulls > 0) ? make_bitview(indices->validity, indices->validity_bit_offset) : none;
  const BitView vvalid = values_nullable ? make_bitview(values->validity, values->validity_bit_offset) : none;
  uint8_t* out_valid = nullptr;
  size_t vbytes = 0;
  if (values_nullable || indices->validity) {
    vbytes = ah_bitmap_bytes(n);
    AH_TRY(ah_out_alloc(ctx, vbytes, (void**)&out_valid));
  }
  // tmp: starts | local offsets | round totals | round bases | {first OOB, valid rows, byte total, -} | tile-scan slots
  const int64_t nrounds = ah_ceil_div(n, 1024);
  const size_t col = ((size_t)n * ow + 7) & ~(size_t)7;
  char* tmp = nullptr;
  ah_status st = ah_pool_alloc(ctx, 2 * col + (size_t)(2 * nrounds + 4 + AH_SCAN_CHAIN_MAX_BLOCKS) * 8, (void**)&tmp);
  if (st != AH_OK) {
    ah_out_free(ctx, out_valid, vbytes);
    return st;
  }
  void* starts = tmp;
  void* loc = tmp + col;
  unsigned long long* round_total = (unsigned long long*)(tmp + 2 * col);
  unsigned long long* round_base = round_total + nrounds;
  unsigned long long* counters = round_base + nrounds;
  unsigned long long* scan_slots = counters + 4;
  hipMemsetAsync(counters, 0xFF, 8, ctx->stream);
  hipMemsetAsync(counters + 1, 0, (size_t)(3 + AH_SCAN_CHAIN_MAX_BLOCKS) * 8, ctx->stream);
  {
    ah_prof_scope ps(ctx, "string_take_ranges");
    st = large ? launch_take_ranges<int64_t>(ctx, values, indices, ivalid, vvalid, (unsigned long long*)out_valid,
                                             (int64_t*)starts, (int64_t*)loc, round_total, counters)
               : launch_take_ranges<int32_t>(ctx, values, indices, ivalid, vvalid, (unsigned long long*)out_valid,
                                             (int32_t*)starts, (int32_t*)loc, round_total, counters);
  }
  if (st == AH_OK) {
    ah_prof_scope ps(ctx, "string_ranges_scan");
    launch_string_tile_scan(ctx, round_total, nrounds, round_base, counters + 2, nullptr, scan_slots);
  }
  if (st == AH_OK) {  // the one wait before the gather: first OOB position, valid rows, byte total
    hipError_t e = ah_d2h_wait(ctx, ctx->pinned + 16, counters, 24);
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "take ranges failed: %s", hipGetErrorString(e));
  }
  const int64_t out_nulls = st == AH_OK ? n - (int64_t)ctx->pinned[17] : 0;
  const uint64_t total_bytes = st == AH_OK ? ctx->pinned[18] : 0;
  if (st == AH_OK && ctx->pinned[16] != ~0ull) {
    if (values_nullable) {
      // take_nulls runs first in the reference: take_bits -> BooleanBuffer::value asserts (boolean.rs:495)
      st = ah_fail(ctx, AH_PANIC, "assertion failed: idx < self.bit_len");
    } else {
      // `input_offsets[index]` / `input_offsets[index + 1]` bounds panic (take.rs:518-519): the slice
      // has values.len()+1 entries, so index == len passes the first access and fails the second
      int w = ah_type_width(indices->type);
      uint64_t raw = 0;
      hipMemcpy(&raw, (const char*)indices->values + (int64_t)ctx->pinned[16] * w, w, hipMemcpyDeviceToHost);
      uint64_t ix;
      switch (indices->type) {
        case AH_INT8: ix = (uint32_t)(int32_t)(int8_t)raw; break;
        case AH_INT16: ix = (uint32_t)(int32_t)(int16_t)raw; break;
        case AH_INT32: ix = (uint32_t)raw; break;
        default: ix = raw; break;
      }
      uint64_t bad = ix == (uint64_t)values->length ? ix + 1 : ix;
      st = ah_fail(ctx, AH_PANIC, "index out of bounds: the len is %lld but the index is %llu",
                   (long long)values->length + 1, (unsigned long long)bad);
    }
  }
  // take_bytes: T::Offset::from_usize(capacity).ok_or_else(OffsetOverflowError(capacity)) (take.rs:521)
  if (st == AH_OK && !large && total_bytes > (uint64_t)INT32_MAX)
    st = ah_fail(ctx, AH_OFFSET_OVERFLOW_ERROR, "%llu", (unsigned long long)total_bytes);
  const size_t ob = (size_t)(n + 1) * ow;
  void *offs = nullptr, *data = nullptr;
  if (st == AH_OK) st = ah_out_alloc(ctx, ob, &offs);
  if (st == AH_OK) st = ah_out_alloc(ctx, (size_t)total_bytes, &data);
  if (st == AH_OK) {
    ah_prof_scope ps(ctx, "string_gather_bytes");
    const unsigned grid = (unsigned)ah_ceil_div(n, 256);
    if (large)
      take_gather_rows_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const uint8_t*)values->values, (const int64_t*)starts, (const int64_t*)loc,
                                                                     round_base, counters + 2, n, (int64_t*)offs, (uint8_t*)data);
    else
      take_gather_rows_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const uint8_t*)values->values, (const int32_t*)starts, (const int32_t*)loc,
                                                                     round_base, counters + 2, n, (int32_t*)offs, (uint8_t*)data);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "string gather failed: %s", hipGetErrorString(e));
  }
  ah_pool_free(ctx, tmp);
  if (st != AH_OK) {
    ah_out_free(ctx, offs, ob);
    ah_out_free(ctx, data, (size_t)total_bytes);
    ah_out_free(ctx, out_valid, vbytes);
    return st;
  }
  out->offsets = offs;
  out->offsets_bytes = (int64_t)ob;
  out->values = data;
  out->values_bytes = (int64_t)total_bytes;
  out->length = n;
  if (out_valid && (out_nulls > 0 || !values_nullable)) {  // take_nulls drops an all-valid result; index nulls are cloned
    out->validity = out_valid;
    out->validity_bytes = (int64_t)vbytes;
    out->null_count = out_nulls;
  } else {
    ah_out_free(ctx, out_valid, vbytes);
  }
  return AH_OK;
}
