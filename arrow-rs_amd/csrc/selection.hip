// parquet `RowSelection`, bitmap-backed form, on the device.
//
// Reference: parquet/src/arrow/arrow_reader/selection/ — mod.rs (`RowSelection::{from_filters :311, and_then
// :462, intersection :482, union :505, offset :566, limit :585, split_off :408, trim :535, iter :612}`),
// algebra.rs (`and_then_masks` :392-436, `intersect_masks` / `union_masks` :267-349: the longer side's tail
// passes through), boolean.rs (`mask_to_selectors` :172-190, `offset_mask` :291-301, `limit_mask` :303-309,
// `trim_mask` :281-289), arrow-buffer/src/buffer/boolean.rs:445 `find_nth_set_bit_position`.
// It is the largest in-tree consumer of `filter`: the row-filter loop of
// arrow_reader/read_plan.rs turns predicate results into a selection and chains them with `and_then`.
//
// Every operation is a pass over bitmaps (1 bit/row, HBM-bound, tiny next to the columns they steer):
//   and_then   : out[i] = mask[i] & other[rank_mask(i)] — a bit "deposit".  Workgroup prefix popcounts (count +
//                scan kernels), then per 64-bit mask word the 64 `other` bits starting at the word's rank are
//                fetched with a funnel shift and deposited with mbcnt (rank below the lane) + __ballot.
//   boundaries : positions where bit[i] != bit[i-1] (bit[-1] = 0) = the run starts of the RLE form; an edge
//                bitmap (x ^ (x << 1 | carry)) followed by set-bit index extraction.
//   from_boundaries : RLE -> bitmap; each output word binary-searches its first boundary and toggles.
//   find_nth_set_bit : prefix table + one-word-at-a-time walk inside the owning tile.
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.hpp"

namespace {

constexpr int SEL_BLOCK = 256;
constexpr int SEL_ITERS = 16;
constexpr int64_t SEL_TILE_WORDS = (int64_t)SEL_BLOCK * SEL_ITERS;  // 4096 words = 262144 rows per workgroup

__device__ __forceinline__ int block_excl_scan(int v, int* total, int* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_scan_incl(v);
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SEL_BLOCK / 64; ++w) {
    if (w < wave) base += sm[w];
    tot += sm[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(SEL_BLOCK) void sel_count_kernel(BitView m, int64_t len, int64_t nwords,
                                                             unsigned long long* counts) {
  unsigned long long c = 0;
  const int64_t w0 = (int64_t)blockIdx.x * SEL_TILE_WORDS;
  for (int it = 0; it < SEL_ITERS; ++it) {
    const int64_t w = w0 + (int64_t)it * SEL_BLOCK + threadIdx.x;
    if (w < nwords) c += __popcll(bv_fetch64(m, w * 64, len));
  }
  c = wave_reduce_add64(c);
  __shared__ unsigned long long sm[SEL_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// prefix[i] = sum(counts[0..i)), prefix[n] = total.  One workgroup; n is a few thousand at 1e9 rows.
__global__ __launch_bounds__(SEL_BLOCK) void sel_scan_kernel(const unsigned long long* counts, int n,
                                                            unsigned long long* prefix) {
  __shared__ unsigned long long part[SEL_BLOCK];
  const int per = (n + SEL_BLOCK - 1) / SEL_BLOCK;
  const int b = threadIdx.x * per, e = min(n, b + per);
  unsigned long long s = 0;
  for (int i = b; i < e; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long acc = 0;
    for (int i = 0; i < SEL_BLOCK; ++i) {
      const unsigned long long t = part[i];
      part[i] = acc;
      acc += t;
    }
    prefix[n] = acc;
  }
  __syncthreads();
  unsigned long long acc = part[threadIdx.x];
  for (int i = b; i < e; ++i) {
    prefix[i] = acc;
    acc += counts[i];
  }
}

__global__ __launch_bounds__(SEL_BLOCK) void sel_expand_kernel(BitView m, int64_t len, int64_t nwords, BitView other,
                                                              int64_t other_len, const unsigned long long* prefix,
                                                              unsigned long long* out) {
  __shared__ int sm[SEL_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  long long base = (long long)prefix[blockIdx.x];
  const int64_t w0 = (int64_t)blockIdx.x * SEL_TILE_WORDS;
  for (int it = 0; it < SEL_ITERS; ++it) {
    const int64_t wbase = w0 + (int64_t)it * SEL_BLOCK;
    if (wbase >= nwords) break;
    const int64_t w = wbase + threadIdx.x;
    const unsigned long long x = w < nwords ? bv_fetch64(m, w * 64, len) : 0ull;
    int total;
    const int rank0 = block_excl_scan(__popcll(x), &total, sm);
    // the 64 bits of `other` this word may consume start at its rank
    const unsigned long long o = x ? bv_fetch64(other, base + rank0, other_len) : 0ull;
    unsigned long long res = 0;
    // deposit: for word j of this wave, lane l contributes bit l
    for (int j = 0; j < 64; ++j) {
      const unsigned long long xj = __shfl(x, j, 64);
      if (xj == 0) continue;  // wave-uniform
      const unsigned long long oj = __shfl(o, j, 64);
      const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(xj >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)xj, 0u));
      const bool bit = ((xj >> lane) & 1ull) && ((oj >> rank) & 1ull);
      const unsigned long long r = __ballot(bit);
      if (lane == j) res = r;
    }
    if (w < nwords) out[w] = res;
    base += total;
  }
}

// intersect_masks / union_masks (algebra.rs:267-349): `op` over the common prefix [0, k), the longer side's tail
// passes through unchanged
__global__ __launch_bounds__(SEL_BLOCK) void sel_combine_kernel(BitView longer, int64_t len, BitView shorter, int64_t k,
                                                               int op, int64_t nwords, unsigned long long* out) {
  const int64_t w = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
  if (w >= nwords) return;
  const unsigned long long x = bv_fetch64(longer, w * 64, len);
  const unsigned long long y = bv_fetch64(shorter, w * 64, k);  // bits at or past k read as 0
  const int64_t in_prefix = k - w * 64;
  const unsigned long long beyond = in_prefix >= 64 ? 0ull : in_prefix <= 0 ? ~0ull : ~((1ull << in_prefix) - 1);
  out[w] = op == 0 ? (x & (y | beyond)) : (x | y);
}

// e[i] = bit[i] ^ bit[i-1] (bit[-1] = 0), positions >= len cleared
__global__ __launch_bounds__(SEL_BLOCK) void sel_edges_kernel(BitView m, int64_t len, int64_t nwords,
                                                             unsigned long long* out) {
  const int64_t w = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
  if (w >= nwords) return;
  const unsigned long long x = bv_fetch64(m, w * 64, len);
  const unsigned long long y = w == 0 ? (x << 1) : bv_fetch64(m, w * 64 - 1, len);
  unsigned long long e = x ^ y;
  const int64_t rem = len - w * 64;
  if (rem < 64) e &= (1ull << rem) - 1;
  out[w] = e;
}

__global__ __launch_bounds__(SEL_BLOCK) void sel_indices_kernel(const unsigned long long* bits, int64_t nwords,
                                                               const unsigned long long* prefix, int64_t* out) {
  __shared__ int sm[SEL_BLOCK / 64];
  long long base = (long long)prefix[blockIdx.x];
  const int64_t w0 = (int64_t)blockIdx.x * SEL_TILE_WORDS;
  for (int it = 0; it < SEL_ITERS; ++it) {
    const int64_t wbase = w0 + (int64_t)it * SEL_BLOCK;
    if (wbase >= nwords) break;
    const int64_t w = wbase + threadIdx.x;
    unsigned long long x = w < nwords ? bits[w] : 0ull;
    int total;
    long long at = base + block_excl_scan(__popcll(x), &total, sm);
    while (x) {
      out[at++] = w * 64 + __builtin_ctzll(x);
      x &= x - 1;
    }
    base += total;
  }
}

// bit p = parity of #{boundaries <= p}
__global__ __launch_bounds__(SEL_BLOCK) void sel_from_bounds_kernel(const int64_t* bounds, int64_t nb, int64_t len,
                                                                   int64_t nwords, unsigned long long* out) {
  const int64_t w = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
  if (w >= nwords) return;
  const int64_t s = w * 64;
  int64_t lo = 0, hi = nb;  // first boundary >= s
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (bounds[mid] < s) lo = mid + 1;
    else hi = mid;
  }
  unsigned long long word = (lo & 1) ? ~0ull : 0ull;
  for (int64_t i = lo; i < nb; ++i) {
    const int64_t b = bounds[i];
    if (b >= s + 64) break;
    word ^= ~0ull << (b - s);
  }
  const int64_t rem = len - s;
  if (rem < 64) word &= (1ull << rem) - 1;
  out[w] = word;
}

// position one past the n-th (1-based) set bit at or after `start`, or len
__global__ void sel_find_nth_kernel(BitView m, int64_t len, int64_t nwords, const unsigned long long* prefix,
                                    int nblocks, int64_t start, unsigned long long n, int64_t* out) {
  // set bits before `start`
  const int64_t sw = start >> 6;
  const int sb = (int)(sw / SEL_TILE_WORDS);
  unsigned long long before = sb < nblocks ? prefix[sb] : prefix[nblocks];
  for (int64_t w = (int64_t)sb * SEL_TILE_WORDS; w < sw && w < nwords; ++w) before += __popcll(bv_fetch64(m, w * 64, len));
  if (sw < nwords && (start & 63)) before += __popcll(bv_fetch64(m, sw * 64, len) & ((1ull << (start & 63)) - 1));
  const unsigned long long target = before + n;  // global 1-based ordinal
  if (target > prefix[nblocks]) {
    *out = len;
    return;
  }
  int lo = 0, hi = nblocks - 1;  // last block with prefix < target
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] < target) lo = mid;
    else hi = mid - 1;
  }
  unsigned long long need = target - prefix[lo];
  for (int64_t w = (int64_t)lo * SEL_TILE_WORDS; w < nwords; ++w) {
    unsigned long long x = bv_fetch64(m, w * 64, len);
    const unsigned long long c = __popcll(x);
    if (c < need) {
      need -= c;
      continue;
    }
    while (--need) x &= x - 1;
    *out = w * 64 + __builtin_ctzll(x) + 1;
    return;
  }
  *out = len;
}

struct Prefix {
  unsigned long long* counts = nullptr;
  unsigned long long* prefix = nullptr;
  int nblocks = 0;
  int64_t total = 0;
};

ah_status build_prefix(ah_context* ctx, BitView m, int64_t len, Prefix* p) {
  const int64_t nwords = ah_ceil_div(len, 64);
  p->nblocks = (int)std::max<int64_t>(1, ah_ceil_div(nwords, SEL_TILE_WORDS));
  void* mem = nullptr;
  AH_TRY(ah_pool_alloc(ctx, sizeof(unsigned long long) * (2 * (size_t)p->nblocks + 2), &mem));
  p->counts = (unsigned long long*)mem;
  p->prefix = p->counts + p->nblocks;
  hipLaunchKernelGGL(sel_count_kernel, dim3(p->nblocks), dim3(SEL_BLOCK), 0, ctx->stream, m, len, nwords, p->counts);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(SEL_BLOCK), 0, ctx->stream, p->counts, p->nblocks, p->prefix);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, p->prefix + p->nblocks, 8);
  if (e != hipSuccess) {
    ah_pool_free(ctx, mem);
    p->counts = nullptr;
    return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in selection prefix", hipGetErrorString(e));
  }
  p->total = (int64_t)ctx->pinned[0];
  return AH_OK;
}

ah_status check_mask(ah_context* ctx, const ah_array_view* v) {
  if (v->type != AH_BOOL) return ah_fail(ctx, AH_INVALID_ARGUMENT, "a row selection mask must be a BooleanArray, got %s", ah_type_name(v->type));
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, v, &nulls));
  if (nulls != 0)  // `assert_eq!(filter.null_count(), 0)` (mod.rs:318)
    return ah_fail(ctx, AH_PANIC, "assertion `left == right` failed\n  left: %lld\n right: 0", (long long)nulls);
  return AH_OK;
}

ah_status alloc_mask_out(ah_context* ctx, int64_t len, ah_array_out* out) {
  out->type = AH_BOOL;
  out->length = len;
  if (len == 0) return AH_OK;
  void* p = nullptr;
  const size_t bytes = ah_bitmap_bytes(len);
  AH_TRY(ah_out_alloc(ctx, bytes, &p));
  out->values = p;
  out->values_bytes = (int64_t)bytes;
  return AH_OK;
}

}  // namespace

extern "C" ah_status ah_selection_and_then(ah_context* ctx, const ah_array_view* mask, const ah_array_view* other,
                                           ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !mask || !other || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  AH_TRY(check_mask(ctx, mask));
  AH_TRY(check_mask(ctx, other));
  const int64_t len = mask->length;
  const BitView m = make_bitview(mask->values, mask->values_bit_offset);
  const BitView o = make_bitview(other->values, other->values_bit_offset);
  Prefix p;
  if (len > 0) AH_TRY(build_prefix(ctx, m, len, &p));
  auto done = [&](ah_status st) {
    if (p.counts) ah_pool_free(ctx, p.counts);
    if (st != AH_OK) ah_array_release(ctx, out);
    return st;
  };
  if (other->length < p.total) return done(ah_fail(ctx, AH_PANIC, "selection contains less than the number of selected rows"));
  if (other->length > p.total) return done(ah_fail(ctx, AH_PANIC, "selection exceeds the number of selected rows"));
  int64_t other_true = 0;
  if (other->length > 0) {
    ah_status st = ah_count_set_bits(ctx, (const uint8_t*)other->values, other->values_bit_offset, other->length, &other_true);
    if (st != AH_OK) return done(st);
  }
  if (other_true == p.total && len > 0) {  // `return mask.clone()` (algebra.rs:404-406): zero-copy
    out->type = AH_BOOL;
    out->length = len;
    out->values = const_cast<void*>(mask->values);
    out->values_bit_offset = mask->values_bit_offset;
    out->values_bytes = (int64_t)ah_ceil_div(mask->values_bit_offset + len, 8);
    out->flags = AH_OUT_BORROWED;
    return done(AH_OK);
  }
  ah_status st = alloc_mask_out(ctx, len, out);
  if (st != AH_OK || len == 0) return done(st);
  const int64_t nwords = ah_ceil_div(len, 64);
  if (other_true == 0) {  // `BooleanBuffer::new_unset(mask.len())` (:401-403)
    hipError_t e = hipMemsetAsync(out->values, 0, (size_t)out->values_bytes, ctx->stream);
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    return done(e == hipSuccess ? AH_OK : ah_fail(ctx, AH_HIP_ERROR, "HIP error %s", hipGetErrorString(e)));
  }
  {
    ah_prof_scope ps(ctx, "selection_and_then");
    hipLaunchKernelGGL(sel_expand_kernel, dim3(p.nblocks), dim3(SEL_BLOCK), 0, ctx->stream, m, len, nwords, o,
                       other->length, p.prefix, (unsigned long long*)out->values);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  return done(e == hipSuccess ? AH_OK : ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in selection and_then", hipGetErrorString(e)));
}

extern "C" ah_status ah_selection_combine(ah_context* ctx, int32_t op, const ah_array_view* l, const ah_array_view* r,
                                          ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !l || !r || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op != 0 && op != 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown selection op %d", op);
  AH_TRY(check_mask(ctx, l));
  AH_TRY(check_mask(ctx, r));
  const ah_array_view* longer = l->length > r->length ? l : r;
  const ah_array_view* shorter = longer == l ? r : l;
  AH_TRY(alloc_mask_out(ctx, longer->length, out));
  if (longer->length == 0) return AH_OK;
  const int64_t nwords = ah_ceil_div(longer->length, 64);
  hipLaunchKernelGGL(sel_combine_kernel, dim3((unsigned)ah_ceil_div(nwords, SEL_BLOCK)), dim3(SEL_BLOCK), 0, ctx->stream,
                     make_bitview(longer->values, longer->values_bit_offset), longer->length,
                     make_bitview(shorter->values, shorter->values_bit_offset), shorter->length, op, nwords,
                     (unsigned long long*)out->values);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  if (e != hipSuccess) {
    ah_array_release(ctx, out);
    return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in selection combine", hipGetErrorString(e));
  }
  return AH_OK;
}

extern "C" ah_status ah_selection_boundaries(ah_context* ctx, const ah_array_view* mask, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !mask || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  AH_TRY(check_mask(ctx, mask));
  out->type = AH_INT64;
  const int64_t len = mask->length;
  if (len == 0) return AH_OK;
  const int64_t nwords = ah_ceil_div(len, 64);
  void* edges = nullptr;
  AH_TRY(ah_pool_alloc(ctx, (size_t)nwords * 8, &edges));
  const BitView m = make_bitview(mask->values, mask->values_bit_offset);
  hipLaunchKernelGGL(sel_edges_kernel, dim3((unsigned)ah_ceil_div(nwords, SEL_BLOCK)), dim3(SEL_BLOCK), 0, ctx->stream, m,
                     len, nwords, (unsigned long long*)edges);
  Prefix p;
  ah_status st = build_prefix(ctx, make_bitview(edges, 0), len, &p);
  if (st == AH_OK && p.total > 0) {
    void* idx = nullptr;
    st = ah_out_alloc(ctx, (size_t)p.total * 8, &idx);
    if (st == AH_OK) {
      out->values = idx;
      out->values_bytes = p.total * 8;
      out->length = p.total;
      {
        ah_prof_scope ps(ctx, "selection_boundaries");
        hipLaunchKernelGGL(sel_indices_kernel, dim3(p.nblocks), dim3(SEL_BLOCK), 0, ctx->stream,
                           (const unsigned long long*)edges, nwords, p.prefix, (int64_t*)idx);
      }
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = ah_stream_wait(ctx);
      if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in selection boundaries", hipGetErrorString(e));
    }
  }
  if (p.counts) ah_pool_free(ctx, p.counts);
  ah_pool_free(ctx, edges);
  if (st != AH_OK) ah_array_release(ctx, out);
  return st;
}

extern "C" ah_status ah_selection_from_boundaries(ah_context* ctx, const ah_array_view* bounds, int64_t total_rows,
                                                  ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !bounds || !out || total_rows < 0) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (bounds->type != AH_INT64) return ah_fail(ctx, AH_INVALID_ARGUMENT, "run boundaries must be Int64, got %s", ah_type_name(bounds->type));
  AH_TRY(alloc_mask_out(ctx, total_rows, out));
  if (total_rows == 0) return AH_OK;
  const int64_t nwords = ah_ceil_div(total_rows, 64);
  hipLaunchKernelGGL(sel_from_bounds_kernel, dim3((unsigned)ah_ceil_div(nwords, SEL_BLOCK)), dim3(SEL_BLOCK), 0, ctx->stream,
                     (const int64_t*)bounds->values, bounds->length, total_rows, nwords, (unsigned long long*)out->values);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_stream_wait(ctx);
  if (e != hipSuccess) {
    ah_array_release(ctx, out);
    return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in selection from_boundaries", hipGetErrorString(e));
  }
  return AH_OK;
}

extern "C" ah_status ah_selection_find_nth_set_bit(ah_context* ctx, const ah_array_view* mask, int64_t start, int64_t n,
                                                   int64_t* pos) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !mask || !pos || start < 0 || n < 0) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  AH_TRY(check_mask(ctx, mask));
  const int64_t len = mask->length;
  if (n == 0) {  // boolean.rs:446-448
    *pos = start;
    return AH_OK;
  }
  if (start >= len) {
    *pos = len;
    return AH_OK;
  }
  const BitView m = make_bitview(mask->values, mask->values_bit_offset);
  Prefix p;
  AH_TRY(build_prefix(ctx, m, len, &p));
  int64_t* dpos = (int64_t*)(p.prefix + p.nblocks + 1);
  hipLaunchKernelGGL(sel_find_nth_kernel, dim3(1), dim3(1), 0, ctx->stream, m, len, ah_ceil_div(len, 64), p.prefix, p.nblocks,
                     start, (unsigned long long)n, dpos);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, dpos, 8);
  ah_pool_free(ctx, p.counts);
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in find_nth_set_bit", hipGetErrorString(e));
  *pos = (int64_t)ctx->pinned[0];
  return AH_OK;
}
