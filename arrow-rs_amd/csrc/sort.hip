// arrow_ord::sort::sort_to_indices on MI355X — the producer of the indices `take` consumes.
//
// Reference: arrow-ord/src/sort.rs — `sort_to_indices` :276-330, `partition_validity` :193-255 (valid and null
// row numbers, each ascending), `sort_primitive` :341-352 / `sort_boolean` :325-339 (pairs (index, value) ordered
// by `T::Native::compare`, i.e. IEEE totalOrder for floats), `sort_impl` :639-672 (descending = reversed
// comparator; nulls first or last, null rows in ascending row order; `limit` keeps a prefix).
// The reference sorts with `sort_unstable_by`: the order of equal keys is unspecified there.  This kernel is a
// STABLE least-significant-digit radix sort, so equal keys stay in ascending row order (also under `descending`),
// which is what every tie in the reference's own tests shows (sort.rs:1625-1895).
//
// Pipeline (all in HBM):
//   1. valid / null row numbers: iota filtered by the validity bitmap and by its complement (filter kernels);
//   2. keys: every valid value mapped to an unsigned integer whose order is the requested order (sign flip for
//      signed ints; floats: negative -> ~bits, positive -> bits ^ sign; descending -> ~key);
//   3. one pass computes all digit histograms (they do not depend on the order), so passes whose digit is the
//      same for every key are skipped outright (an Int64 column of small values sorts in 3 passes, not 8);
//   4. per remaining 8-bit digit: per-workgroup histogram -> exclusive scan -> stable scatter.  The scatter ranks
//      keys inside a wave with 8 ballots ("which lanes hold my digit") + mbcnt, stages the tile digit-sorted in
//      LDS and writes runs of equal digits contiguously.
// HBM traffic per pass: keys read twice, (key, index) pairs written once: 8+12+12 = 32 B/row for 64-bit keys.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "common.hpp"

namespace {

constexpr int RS_BLOCK = 256;
#ifndef RS_ITER_N
#define RS_ITER_N 16
#endif
constexpr int RS_ITER = RS_ITER_N;
constexpr int RS_TILE = RS_BLOCK * RS_ITER;  // 4096 pairs per tile
constexpr int RS_MAX_BLOCKS = 2048;

// exclusive scan of one counter per thread across the 256-thread workgroup (unsigned: totals reach 2^32 - 1 pairs)
__device__ __forceinline__ unsigned wave_scan_incl_u(unsigned v) {
  const unsigned lane = __lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = (unsigned)__shfl_up((int)v, o, 64);
    if (lane >= (unsigned)o) v += t;
  }
  return v;
}
__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned* total, unsigned* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned incl = wave_scan_incl_u(v);
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RS_BLOCK / 64; ++w) {
    if (w < wave) base += sm[w];
    tot += sm[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// raw W-byte value (zero-extended) -> unsigned key in the requested order; mode 0 unsigned, 1 signed, 2 float
template <int W, typename KT> __device__ __forceinline__ KT order_key(KT b, int mode, bool desc) {
  constexpr KT sign = (KT)1 << (W * 8 - 1);
  constexpr KT mask = (W == 8 || W == 4) ? ~(KT)0 : (((KT)1 << (W * 8 - 1)) << 1) - 1;
  KT k = mode == 0 ? b : mode == 1 ? (b ^ sign) : ((b & sign) ? (~b & mask) : (b ^ sign));
  return desc ? (~k & mask) : k;
}

// `nulls_to_zero` (lexsort): null rows get one constant key, so the stable key passes leave them in the order the
// later columns gave them; the 1-bit null pass then moves them as a block
template <int W, typename KT>
__global__ void sort_keys_kernel(const void* values, const uint32_t* rows, int64_t m, int mode, int desc,
                                 KT* keys, uint32_t* idx, BitView nulls_to_zero = BitView{nullptr, 0}) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t r = rows ? rows[i] : (uint32_t)i;
  if (nulls_to_zero.words && !bv_get(nulls_to_zero, r)) {
    keys[i] = 0;
    return;
  }
  KT raw;
  if constexpr (W == 8) raw = ((const uint64_t*)values)[r];
  else if constexpr (W == 4) raw = ((const uint32_t*)values)[r];
  else if constexpr (W == 2) raw = ((const uint16_t*)values)[r];
  else raw = ((const uint8_t*)values)[r];
  keys[i] = order_key<W, KT>(raw, mode, desc != 0);
  if (!rows) idx[i] = r;  // with nulls the row numbers ARE the initial index column
}

// Boolean values: key = bit (or its complement)
__global__ void sort_bool_keys_kernel(BitView bits, int64_t len, const uint32_t* rows, int64_t m, int desc,
                                      uint32_t* keys, uint32_t* idx, BitView nulls_to_zero = BitView{nullptr, 0}) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t r = rows ? rows[i] : (uint32_t)i;
  if (nulls_to_zero.words && !bv_get(nulls_to_zero, r)) {
    keys[i] = 0;
    return;
  }
  keys[i] = (uint32_t)bv_get(bits, r) ^ (desc ? 1u : 0u);
  if (!rows) idx[i] = r;
}

// lexsort: digit 0 / 1 that puts the null rows of a column first or last (rows taken in the current order)
__global__ void sort_null_keys_kernel(BitView valid, const uint32_t* rows, int64_t m, int nulls_first, uint32_t* keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned is_null = bv_get(valid, rows[i]) ? 0u : 1u;
  keys[i] = nulls_first ? (is_null ^ 1u) : is_null;
}

__global__ void iota_u32_kernel(uint32_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i;
}

// ---- Utf8 / LargeUtf8 (sort_bytes, arrow-ord/src/sort.rs: `a.cmp(b)` on &[u8] = bytewise lexicographic, a proper
// prefix first).  A string column is sorted as a chain of fixed-width key columns, least significant first:
// the byte length, then bytes [8j, 8j+8) as one big-endian u64 (zero padded) for j = last .. 0.  Zero padding makes
// a proper prefix compare <= its extensions on every chunk and the length column breaks exactly those ties
// (the extension is longer), so the chain reproduces the bytewise order, NUL bytes included.
template <typename OFF>
__global__ void string_max_len_kernel(const OFF* offs, int64_t n, unsigned long long* out) {
  unsigned long long mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long l = (unsigned long long)(offs[i + 1] - offs[i]);
    mx = l > mx ? l : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(mx, o, 64);
    mx = other > mx ? other : mx;
  }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}

template <typename OFF>
__global__ void string_len_keys_kernel(const OFF* offs, const uint32_t* rows, int64_t m, int desc, uint64_t* keys, BitView nulls_to_zero) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t r = rows[i];
  if (nulls_to_zero.words && !bv_get(nulls_to_zero, r)) {
    keys[i] = 0;
    return;
  }
  const uint64_t l = (uint64_t)(offs[r + 1] - offs[r]);
  keys[i] = desc ? ~l : l;
}

template <typename OFF>
__global__ void string_chunk_keys_kernel(const OFF* offs, const uint8_t* data, const uint32_t* rows, int64_t m, int64_t byte0,
                                         int desc, uint64_t* keys, BitView nulls_to_zero) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t r = rows[i];
  if (nulls_to_zero.words && !bv_get(nulls_to_zero, r)) {
    keys[i] = 0;
    return;
  }
  const int64_t a = (int64_t)offs[r], n = (int64_t)offs[r + 1] - a;
  uint64_t k = 0;
  if (byte0 + 8 <= n) {
    uint64_t raw;
    __builtin_memcpy(&raw, data + a + byte0, 8);
    k = __builtin_bswap64(raw);
  } else {
    for (int b = 0; b < 8; ++b) k = (k << 8) | (byte0 + b < n ? (uint64_t)data[a + byte0 + b] : 0ull);
  }
  keys[i] = desc ? ~k : k;
}

// all digit histograms of the key column at once: hist[pass][256]
template <typename KT, int PASSES>
__global__ __launch_bounds__(RS_BLOCK) void rs_digit_census_kernel(const KT* keys, int64_t m, unsigned long long* hist) {
  __shared__ unsigned int sm[PASSES][256];
  for (int i = threadIdx.x; i < PASSES * 256; i += RS_BLOCK) (&sm[0][0])[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x; i < m; i += (int64_t)gridDim.x * RS_BLOCK) {
    const KT k = keys[i];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) atomicAdd(&sm[p][(k >> (8 * p)) & 255], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PASSES * 256; i += RS_BLOCK) {
    const unsigned int c = (&sm[0][0])[i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}

// per-workgroup histogram of one digit over the workgroup's contiguous run of tiles: out[d * nblocks + b]
template <typename KT>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const KT* keys, int64_t m, int shift, int64_t tiles_per_block,
                                                          unsigned int* out) {
  __shared__ unsigned int sm[256];
  sm[threadIdx.x] = 0;
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * tiles_per_block * RS_TILE;
  const int64_t end = min(m, begin + tiles_per_block * RS_TILE);
  for (int64_t i = begin + threadIdx.x; i < end; i += RS_BLOCK) atomicAdd(&sm[(keys[i] >> shift) & 255], 1u);
  __syncthreads();
  out[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sm[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's per-workgroup counts in place + the digit total.
// (A single workgroup scanning all 256 * nblocks counters took 0.77 ms per pass.)
__global__ __launch_bounds__(RS_BLOCK) void rs_scan_kernel(unsigned int* h, int nblocks, unsigned int* digit_total) {
  __shared__ unsigned sm[RS_BLOCK / 64];
  unsigned int* row = h + (size_t)blockIdx.x * nblocks;
  const int per = (nblocks + RS_BLOCK - 1) / RS_BLOCK;
  const int b = threadIdx.x * per, e = min(nblocks, b + per);
  unsigned int s = 0;
  for (int i = b; i < e; ++i) s += row[i];
  unsigned total;
  unsigned int acc = block_excl_scan_256(s, &total, sm);
  for (int i = b; i < e; ++i) {
    const unsigned int t = row[i];
    row[i] = acc;
    acc += t;
  }
  if (threadIdx.x == 0) digit_total[blockIdx.x] = total;
}

template <typename KT>
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const KT* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                             int64_t m, int shift, int64_t tiles_per_block,
                                                             const unsigned int* __restrict__ offsets,
                                                             const unsigned int* __restrict__ digit_total,
                                                             KT* __restrict__ keys_out, uint32_t* __restrict__ idx_out) {
  __shared__ KT skey[RS_TILE];
  __shared__ uint32_t sidx[RS_TILE];
  __shared__ unsigned int wcnt[RS_BLOCK / 64][256];  // per-wave digit counts of the current tile
  __shared__ unsigned int tpre[256];                 // exclusive prefix over digits inside the tile
  __shared__ unsigned int tcnt[256];
  __shared__ unsigned int running[256];              // next global output slot of each digit for this workgroup
  __shared__ unsigned scan_sm[RS_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {  // first output slot of digit d for this workgroup = (pairs with smaller digits) + (digit d in earlier workgroups)
    unsigned total;
    const unsigned dbase = block_excl_scan_256(digit_total[threadIdx.x], &total, scan_sm);
    running[threadIdx.x] = dbase + offsets[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
  }
  const int64_t first_tile = (int64_t)blockIdx.x * tiles_per_block;
  for (int64_t t = 0; t < tiles_per_block; ++t) {
    const int64_t base = (first_tile + t) * RS_TILE;
    if (base >= m) break;
#pragma unroll
    for (int w = 0; w < RS_BLOCK / 64; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    KT key[RS_ITER];
    uint32_t val[RS_ITER];
    unsigned int local[RS_ITER];
    // phase A0: all loads of the tile first, so 2*RS_ITER requests per lane are in flight at once (issuing each
    // load right before its ranking step serialised RS_ITER memory round trips per tile: 4.3 -> 2.9 ms per pass for
    // everything but the stores)
#pragma unroll
    for (int it = 0; it < RS_ITER; ++it) {
      const int64_t p = base + (int64_t)wave * (RS_ITER * 64) + it * 64 + lane;
      const bool active = p < m;
      key[it] = active ? keys_in[p] : (KT)0;
      val[it] = active ? idx_in[p] : 0u;
    }
    // phase A: stable rank of every pair among the pairs of its wave that share its digit
#pragma unroll
    for (int it = 0; it < RS_ITER; ++it) {
      const int64_t p = base + (int64_t)wave * (RS_ITER * 64) + it * 64 + lane;
      const bool active = p < m;
      const unsigned d = (unsigned)((key[it] >> shift) & 255);
      unsigned long long peers = __ballot(active);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long mb = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? mb : ~mb;
      }
      const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
      unsigned basec = 0;
      if (active) basec = wcnt[wave][d];
      __builtin_amdgcn_wave_barrier();
      if (active && below == 0) wcnt[wave][d] = basec + (unsigned)__popcll(peers);  // first lane of each group
      __builtin_amdgcn_wave_barrier();
      local[it] = basec + below;
    }
    __syncthreads();
    // phase B: digit totals of the tile, prefix over digits, per-wave bases
    {
      const int d = threadIdx.x;
      unsigned tot = 0;
#pragma unroll
      for (int w = 0; w < RS_BLOCK / 64; ++w) {
        const unsigned c = wcnt[w][d];
        wcnt[w][d] = tot;  // becomes: pairs with digit d in earlier waves
        tot += c;
      }
      tcnt[d] = tot;
      unsigned total;
      tpre[d] = block_excl_scan_256(tot, &total, scan_sm);
    }
    __syncthreads();
    // phase C: stage the tile digit-sorted in LDS
#pragma unroll
    for (int it = 0; it < RS_ITER; ++it) {
      const int64_t p = base + (int64_t)wave * (RS_ITER * 64) + it * 64 + lane;
      if (p < m) {
        const unsigned d = (unsigned)((key[it] >> shift) & 255);
        const unsigned pos = tpre[d] + wcnt[wave][d] + local[it];
        skey[pos] = key[it];
        sidx[pos] = val[it];
      }
    }
    __syncthreads();
    // phase D: runs of equal digits go out contiguously
    const int64_t in_tile = min((int64_t)RS_TILE, m - base);
    for (int q = threadIdx.x; q < in_tile; q += RS_BLOCK) {
      const KT k = skey[q];
      const unsigned d = (unsigned)((k >> shift) & 255);
      const size_t dst = (size_t)running[d] + (q - tpre[d]);
      keys_out[dst] = k;
      idx_out[dst] = sidx[q];
    }
    __syncthreads();
    running[threadIdx.x] += tcnt[threadIdx.x];
    __syncthreads();
  }
}

struct Scratch {
  ah_context* ctx;
  std::vector<void*> ptrs;
  ~Scratch() {
    for (void* p : ptrs) ah_pool_free(ctx, p);
  }
  ah_status get(size_t bytes, void** out) {
    AH_TRY(ah_pool_alloc(ctx, bytes ? bytes : 8, out));
    ptrs.push_back(*out);
    return AH_OK;
  }
};

// stable LSD radix sort of (keys, idx) pairs; result pointers come back in *keys / *idx
template <typename KT, int PASSES>
ah_status radix_sort_pairs(ah_context* ctx, Scratch& sc, KT** keys, uint32_t** idx, int64_t m) {
  if (m <= 1) return AH_OK;
  KT* kb = nullptr;
  uint32_t* ib = nullptr;
  unsigned long long* census = nullptr;
  unsigned int* hist = nullptr;
  AH_TRY(sc.get((size_t)m * sizeof(KT), (void**)&kb));
  AH_TRY(sc.get((size_t)m * 4, (void**)&ib));
  AH_TRY(sc.get(PASSES * 256 * 8, (void**)&census));
  const int64_t ntiles = ah_ceil_div(m, RS_TILE);
  const int nblocks = (int)std::min<int64_t>(ntiles, RS_MAX_BLOCKS);
  const int64_t tiles_per_block = ah_ceil_div(ntiles, nblocks);
  AH_TRY(sc.get((size_t)256 * (nblocks + 1) * 4, (void**)&hist));
  unsigned int* digit_total = hist + (size_t)256 * nblocks;
  AH_HIP(ctx, hipMemsetAsync(census, 0, PASSES * 256 * 8, ctx->stream));
  hipLaunchKernelGGL((rs_digit_census_kernel<KT, PASSES>), dim3((unsigned)std::min<int64_t>(ah_ceil_div(m, RS_BLOCK), 2048)),
                     dim3(RS_BLOCK), 0, ctx->stream, *keys, m, census);
  std::vector<unsigned long long> host((size_t)PASSES * 256);
  AH_HIP(ctx, hipMemcpyAsync(host.data(), census, PASSES * 256 * 8, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  KT* kin = *keys;
  uint32_t* iin = *idx;
  for (int p = 0; p < PASSES; ++p) {
    bool trivial = false;  // every key has the same digit: the pass would be the identity
    for (int d = 0; d < 256; ++d) trivial |= host[(size_t)p * 256 + d] == (unsigned long long)m;
    if (trivial) continue;
    ah_prof_scope ps(ctx, "sort_radix_pass");
    hipLaunchKernelGGL((rs_hist_kernel<KT>), dim3(nblocks), dim3(RS_BLOCK), 0, ctx->stream, kin, m, 8 * p, tiles_per_block, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(RS_BLOCK), 0, ctx->stream, hist, nblocks, digit_total);
    hipLaunchKernelGGL((rs_scatter_kernel<KT>), dim3(nblocks), dim3(RS_BLOCK), 0, ctx->stream, kin, iin, m, 8 * p,
                       tiles_per_block, hist, digit_total, kb, ib);
    std::swap(kin, kb);
    std::swap(iin, ib);
  }
  AH_HIP(ctx, hipGetLastError());
  *keys = kin;
  *idx = iin;
  return AH_OK;
}

}  // namespace

static ah_status lexsort_chain(ah_context* ctx, int32_t n_cols, const ah_array_view* cols, const int32_t* descending,
                               const int32_t* nulls_first, int64_t limit, ah_array_out* out);

extern "C" ah_status ah_sort_to_indices(ah_context* ctx, const ah_array_view* v, int32_t descending, int32_t nulls_first,
                                        int64_t limit, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  out->type = AH_UINT32;
  const ah_type t = v->type;
  const int64_t n = v->length;
  const int w = ah_type_width(t);
  if (t == AH_UTF8 || t == AH_LARGE_UTF8) {  // sort_bytes: the chained key-column sort the lexsort path implements
    const int32_t d = descending != 0, nf = nulls_first != 0;
    return lexsort_chain(ctx, 1, v, &d, &nf, limit, out);
  }
  const bool ok = t == AH_BOOL || ah_type_is_integer(t) || t == AH_FLOAT16 || t == AH_FLOAT32 || t == AH_FLOAT64;
  if (!ok) return ah_fail(ctx, AH_COMPUTE_ERROR, "Sort not supported for data type %s", ah_type_name(t));  // sort.rs:324
  if (n == 0 || limit == 0) return AH_OK;  // :281-283
  if (n > (int64_t)UINT32_MAX) return ah_fail(ctx, AH_INVALID_ARGUMENT, "sort_to_indices returns UInt32 indices: %lld rows do not fit", (long long)n);
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, v, &nulls));
  const bool has_nulls = v->validity && nulls > 0;
  const int64_t m = n - nulls;
  const int64_t lim = limit < 0 ? n : std::min(limit, n);
  Scratch sc{ctx, {}};
  auto grid = [](int64_t k) { return dim3((unsigned)std::max<int64_t>(1, ah_ceil_div(k, 256))); };

  // 1. partition_validity (:193): ascending valid rows and ascending null rows
  ah_array_out valid_rows, null_rows;
  ah_out_init(&valid_rows);
  ah_out_init(&null_rows);
  struct Rel {
    ah_context* c;
    ah_array_out *a, *b;
    ~Rel() {
      ah_array_release(c, a);
      ah_array_release(c, b);
    }
  } rel{ctx, &valid_rows, &null_rows};
  if (has_nulls) {
    uint32_t* iota = nullptr;
    AH_TRY(sc.get((size_t)n * 4, (void**)&iota));
    hipLaunchKernelGGL(iota_u32_kernel, grid(n), dim3(256), 0, ctx->stream, iota, n);
    ah_array_view rows{}, pred{};
    rows.type = AH_UINT32;
    rows.length = n;
    rows.values = iota;
    pred.type = AH_BOOL;
    pred.length = n;
    pred.values = v->validity;
    pred.values_bit_offset = v->validity_bit_offset;
    AH_TRY(ah_filter(ctx, &rows, &pred, &valid_rows));
    ah_array_out inv;
    AH_TRY(ah_boolean_unary(ctx, AH_BOOL_NOT, &pred, &inv));
    ah_array_view ip{};
    ip.type = AH_BOOL;
    ip.length = n;
    ip.values = inv.values;
    ip.values_bit_offset = inv.values_bit_offset;
    ah_status st = ah_filter(ctx, &rows, &ip, &null_rows);
    ah_array_release(ctx, &inv);
    AH_TRY(st);
    if (valid_rows.length != m || null_rows.length != nulls)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "null_count %lld does not match the validity bitmap", (long long)nulls);
  }
  const uint32_t* rows = has_nulls ? (const uint32_t*)valid_rows.values : nullptr;

  // 2. keys (+ initial index column) of the valid rows, 3./4. stable radix sort
  uint32_t* sorted = nullptr;
  if (m > 0) {
    uint32_t* idx = nullptr;
    if (has_nulls) {
      AH_TRY(sc.get((size_t)m * 4, (void**)&idx));
      AH_HIP(ctx, hipMemcpyAsync(idx, rows, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      AH_TRY(sc.get((size_t)m * 4, (void**)&idx));
    }
    const int mode = (t == AH_FLOAT16 || t == AH_FLOAT32 || t == AH_FLOAT64) ? 2 : ah_type_is_signed(t) ? 1 : 0;
    if (w == 8) {
      uint64_t* keys = nullptr;
      AH_TRY(sc.get((size_t)m * 8, (void**)&keys));
      hipLaunchKernelGGL((sort_keys_kernel<8, uint64_t>), grid(m), dim3(256), 0, ctx->stream, v->values, rows, m, mode,
                         descending, keys, idx);
      AH_TRY((radix_sort_pairs<uint64_t, 8>(ctx, sc, &keys, &idx, m)));
    } else {
      uint32_t* keys = nullptr;
      AH_TRY(sc.get((size_t)m * 4, (void**)&keys));
      if (t == AH_BOOL) {
        hipLaunchKernelGGL(sort_bool_keys_kernel, grid(m), dim3(256), 0, ctx->stream,
                           make_bitview(v->values, v->values_bit_offset), n, rows, m, descending, keys, idx);
        AH_TRY((radix_sort_pairs<uint32_t, 1>(ctx, sc, &keys, &idx, m)));
      } else if (w == 4) {
        hipLaunchKernelGGL((sort_keys_kernel<4, uint32_t>), grid(m), dim3(256), 0, ctx->stream, v->values, rows, m, mode,
                           descending, keys, idx);
        AH_TRY((radix_sort_pairs<uint32_t, 4>(ctx, sc, &keys, &idx, m)));
      } else if (w == 2) {
        hipLaunchKernelGGL((sort_keys_kernel<2, uint32_t>), grid(m), dim3(256), 0, ctx->stream, v->values, rows, m, mode,
                           descending, keys, idx);
        AH_TRY((radix_sort_pairs<uint32_t, 2>(ctx, sc, &keys, &idx, m)));
      } else {
        hipLaunchKernelGGL((sort_keys_kernel<1, uint32_t>), grid(m), dim3(256), 0, ctx->stream, v->values, rows, m, mode,
                           descending, keys, idx);
        AH_TRY((radix_sort_pairs<uint32_t, 1>(ctx, sc, &keys, &idx, m)));
      }
    }
    sorted = idx;
  }

  // 5. sort_impl (:656-671): nulls first or last, then the limit
  void* res = nullptr;
  AH_TRY(ah_out_alloc(ctx, (size_t)lim * 4, &res));
  uint32_t* o = (uint32_t*)res;
  hipError_t e = hipSuccess;
  const int64_t n_first = nulls_first ? nulls : m;
  const uint32_t* first = nulls_first ? (const uint32_t*)null_rows.values : sorted;
  const uint32_t* second = nulls_first ? sorted : (const uint32_t*)null_rows.values;
  const int64_t a = std::min(lim, n_first), b = lim - a;
  if (a > 0) e = hipMemcpyAsync(o, first, (size_t)a * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess && b > 0) e = hipMemcpyAsync(o + a, second, (size_t)b * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  if (e != hipSuccess) {
    ah_out_free(ctx, res, (size_t)lim * 4);
    return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in sort_to_indices", hipGetErrorString(e));
  }
  out->length = lim;
  out->values = res;
  out->values_bytes = lim * 4;
  return AH_OK;
}

namespace {

// stable sort of the row order `*idx` (n rows) by one column: keys gathered in the current order, radix passes,
// then one more stable 1-bit pass that moves the column's null rows to the front / back
ah_status sort_rows_by_column(ah_context* ctx, const ah_array_view* v, bool desc, bool nulls_first, uint32_t* idx_buf,
                              int64_t n) {
  Scratch sc{ctx, {}};  // this column's temporaries go back to the pool when it is done (same stream: safe)
  uint32_t* cur = idx_buf;
  uint32_t** idx = &cur;
  const ah_type t = v->type;
  const int w = ah_type_width(t);
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;
  const bool ok = is_str || t == AH_BOOL || ah_type_is_integer(t) || t == AH_FLOAT16 || t == AH_FLOAT32 || t == AH_FLOAT64;
  if (!ok) return ah_fail(ctx, AH_COMPUTE_ERROR, "Sort not supported for data type %s", ah_type_name(t));
  const dim3 grid((unsigned)std::max<int64_t>(1, ah_ceil_div(n, 256)));
  const int mode = (t == AH_FLOAT16 || t == AH_FLOAT32 || t == AH_FLOAT64) ? 2 : ah_type_is_signed(t) ? 1 : 0;
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, v, &nulls));
  const bool has_nulls = v->validity && nulls > 0;
  const BitView nz = has_nulls ? make_bitview(v->validity, v->validity_bit_offset) : BitView{nullptr, 0};
  if (is_str) {
    if (!v->offsets) return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
    const bool large = t == AH_LARGE_UTF8;
    unsigned long long* dmax = nullptr;
    AH_TRY(sc.get(8, (void**)&dmax));
    AH_HIP(ctx, hipMemsetAsync(dmax, 0, 8, ctx->stream));
    const dim3 rgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(n, 256), 4096)));
    if (large) hipLaunchKernelGGL(string_max_len_kernel<int64_t>, rgrid, dim3(256), 0, ctx->stream, (const int64_t*)v->offsets, n, dmax);
    else hipLaunchKernelGGL(string_max_len_kernel<int32_t>, rgrid, dim3(256), 0, ctx->stream, (const int32_t*)v->offsets, n, dmax);
    unsigned long long max_len = 0;
    AH_HIP(ctx, hipMemcpyAsync(&max_len, dmax, 8, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(ctx, ah_stream_wait(ctx));
    uint64_t* keys = nullptr;
    AH_TRY(sc.get((size_t)n * 8, (void**)&keys));
    const uint8_t* data = (const uint8_t*)v->values;
    {  // least significant: the length
      uint64_t* k = keys;
      if (large) hipLaunchKernelGGL(string_len_keys_kernel<int64_t>, grid, dim3(256), 0, ctx->stream, (const int64_t*)v->offsets, *idx, n, (int)desc, k, nz);
      else hipLaunchKernelGGL(string_len_keys_kernel<int32_t>, grid, dim3(256), 0, ctx->stream, (const int32_t*)v->offsets, *idx, n, (int)desc, k, nz);
      AH_TRY((radix_sort_pairs<uint64_t, 8>(ctx, sc, &k, idx, n)));
    }
    for (int64_t j = (int64_t)((max_len + 7) / 8) - 1; j >= 0; --j) {  // 8-byte chunks, last to first
      uint64_t* k = keys;
      if (large) hipLaunchKernelGGL(string_chunk_keys_kernel<int64_t>, grid, dim3(256), 0, ctx->stream, (const int64_t*)v->offsets, data, *idx, n, j * 8, (int)desc, k, nz);
      else hipLaunchKernelGGL(string_chunk_keys_kernel<int32_t>, grid, dim3(256), 0, ctx->stream, (const int32_t*)v->offsets, data, *idx, n, j * 8, (int)desc, k, nz);
      AH_TRY((radix_sort_pairs<uint64_t, 8>(ctx, sc, &k, idx, n)));
    }
  } else if (w == 8) {
    uint64_t* keys = nullptr;
    AH_TRY(sc.get((size_t)n * 8, (void**)&keys));
    hipLaunchKernelGGL((sort_keys_kernel<8, uint64_t>), grid, dim3(256), 0, ctx->stream, v->values, *idx, n, mode, (int)desc, keys, *idx, nz);
    AH_TRY((radix_sort_pairs<uint64_t, 8>(ctx, sc, &keys, idx, n)));
  } else {
    uint32_t* keys = nullptr;
    AH_TRY(sc.get((size_t)n * 4, (void**)&keys));
    if (t == AH_BOOL) {
      hipLaunchKernelGGL(sort_bool_keys_kernel, grid, dim3(256), 0, ctx->stream, make_bitview(v->values, v->values_bit_offset),
                         n, *idx, n, (int)desc, keys, *idx, nz);
      AH_TRY((radix_sort_pairs<uint32_t, 1>(ctx, sc, &keys, idx, n)));
    } else if (w == 4) {
      hipLaunchKernelGGL((sort_keys_kernel<4, uint32_t>), grid, dim3(256), 0, ctx->stream, v->values, *idx, n, mode, (int)desc, keys, *idx, nz);
      AH_TRY((radix_sort_pairs<uint32_t, 4>(ctx, sc, &keys, idx, n)));
    } else if (w == 2) {
      hipLaunchKernelGGL((sort_keys_kernel<2, uint32_t>), grid, dim3(256), 0, ctx->stream, v->values, *idx, n, mode, (int)desc, keys, *idx, nz);
      AH_TRY((radix_sort_pairs<uint32_t, 2>(ctx, sc, &keys, idx, n)));
    } else {
      hipLaunchKernelGGL((sort_keys_kernel<1, uint32_t>), grid, dim3(256), 0, ctx->stream, v->values, *idx, n, mode, (int)desc, keys, *idx, nz);
      AH_TRY((radix_sort_pairs<uint32_t, 1>(ctx, sc, &keys, idx, n)));
    }
  }
  if (has_nulls) {
    uint32_t* nk = nullptr;
    AH_TRY(sc.get((size_t)n * 4, (void**)&nk));
    hipLaunchKernelGGL(sort_null_keys_kernel, grid, dim3(256), 0, ctx->stream, make_bitview(v->validity, v->validity_bit_offset),
                       *idx, n, (int)nulls_first, nk);
    AH_TRY((radix_sort_pairs<uint32_t, 1>(ctx, sc, &nk, idx, n)));
  }
  if (cur != idx_buf) AH_HIP(ctx, hipMemcpyAsync(idx_buf, cur, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  return AH_OK;
}

}  // namespace

// lexsort_to_indices (arrow-ord/src/sort.rs:939-1020) as a least-significant-COLUMN-first chain of stable sorts:
// the last column orders the rows first, every earlier column re-sorts them stably, so rows equal on a column
// keep the order the later columns gave them.  Rows equal on every column stay in ascending row order (the
// reference's `sort_unstable_by` / top-k heap leave that order open).
extern "C" ah_status ah_lexsort_to_indices(ah_context* ctx, int32_t n_cols, const ah_array_view* cols,
                                           const int32_t* descending, const int32_t* nulls_first, int64_t limit,
                                           ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out || (n_cols > 0 && (!cols || !descending || !nulls_first))) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  out->type = AH_UINT32;
  if (n_cols <= 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "Sort requires at least one column");  // :944-948
  if (n_cols == 1) return ah_sort_to_indices(ctx, &cols[0], descending[0], nulls_first[0], limit, out);  // :949-953
  return lexsort_chain(ctx, n_cols, cols, descending, nulls_first, limit, out);
}

static ah_status lexsort_chain(ah_context* ctx, int32_t n_cols, const ah_array_view* cols, const int32_t* descending,
                               const int32_t* nulls_first, int64_t limit, ah_array_out* out) {
  const int64_t n = cols[0].length;
  for (int c = 1; c < n_cols; ++c)
    if (cols[c].length != n) return ah_fail(ctx, AH_COMPUTE_ERROR, "lexical sort columns have different row counts");
  const int64_t lim = limit < 0 ? n : std::min(limit, n);
  if (lim == 0) return AH_OK;
  if (n > (int64_t)UINT32_MAX) return ah_fail(ctx, AH_INVALID_ARGUMENT, "lexsort_to_indices returns UInt32 indices: %lld rows do not fit", (long long)n);
  Scratch sc{ctx, {}};
  uint32_t* idx = nullptr;
  AH_TRY(sc.get((size_t)n * 4, (void**)&idx));
  hipLaunchKernelGGL(iota_u32_kernel, dim3((unsigned)ah_ceil_div(n, 256)), dim3(256), 0, ctx->stream, idx, n);
  for (int c = n_cols - 1; c >= 0; --c) {
    AH_TRY(sort_rows_by_column(ctx, &cols[c], descending[c] != 0, nulls_first[c] != 0, idx, n));
  }
  void* res = nullptr;
  AH_TRY(ah_out_alloc(ctx, (size_t)lim * 4, &res));
  hipError_t e = hipMemcpyAsync(res, idx, (size_t)lim * 4, hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  if (e != hipSuccess) {
    ah_out_free(ctx, res, (size_t)lim * 4);
    return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in lexsort_to_indices", hipGetErrorString(e));
  }
  out->length = lim;
  out->values = res;
  out->values_bytes = lim * 4;
  return AH_OK;
}
