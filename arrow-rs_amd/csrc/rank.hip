// rank.hip — arrow_ord::rank::rank (arrow-ord/src/rank.rs:58-160): every row's position in the sorted order, ties
// sharing the HIGHEST of their positions, nulls sharing one rank before or after all values.
//
// Reference: primitive_rank :72-87 collects (value, row) of the valid rows, rank_impl :119-160 sorts them (reversed
// when descending), gives the last element `valid_rank` = len (nulls first) or the number of valid rows (nulls last),
// walks backwards lowering the rank by the size of each finished run of `is_eq` values, and pre-fills the output with
// `null_rank` = null count (nulls first) or len (nulls last).  boolean_rank :182-245 is the same function computed
// from three counts.  Equivalently: rank(row) = first_valid_rank_base + (position of the END of the row's run in the
// sorted order) + 1.
//
// MI355X design: composition of kernels that already exist, plus one small one.
//   perm   = sort_to_indices(values, options)             (stable LSD radix sort, sort.hip)
//   sorted = take(values, perm[valid part])               (gather)
//   edge   = neq(sorted[0..m-1], sorted[1..m])            (run ends; float `neq` is bit inequality under totalOrder,
//                                                          which is rank's `is_eq`)
//   ends   = filter(iota, edge)                           (positions of the run ends, ascending)
//   rank_fill: lane p binary-searches `ends` for the first end >= p (neighbouring lanes walk the same path, so the
//              loads broadcast) and scatters base + end + 1 to out[perm[p]]; null rows get the constant.
#include "common.hpp"

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void rank_fill_kernel(const uint32_t* perm, int64_t m, const uint32_t* ends, int64_t n_ends,
                                                        uint32_t base, uint32_t* out) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < m; p += (int64_t)gridDim.x * 256) {
    int64_t lo = 0, hi = n_ends;  // first j with ends[j] >= p
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)ends[mid] < p) lo = mid + 1;
      else hi = mid;
    }
    const int64_t end = lo < n_ends ? (int64_t)ends[lo] : m - 1;  // the last run ends at the last valid row
    out[perm[p]] = base + (uint32_t)end + 1u;
  }
}

__global__ __launch_bounds__(256) void rank_const_kernel(const uint32_t* perm, int64_t n, uint32_t value, uint32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[perm[i]] = value;
}

struct Temp {  // a library-owned intermediate result, released on every exit path
  ah_context* ctx;
  ah_array_out out;
  explicit Temp(ah_context* c) : ctx(c) { ah_out_init(&out); }
  ~Temp() { ah_array_release(ctx, &out); }
};

ah_array_view view_of(const ah_array_out& o) {
  ah_array_view v{};
  v.type = o.type;
  v.length = o.length;
  v.null_count = 0;
  v.values = o.values;
  v.values_bit_offset = o.values_bit_offset;
  v.offsets = o.offsets;
  return v;
}

}  // namespace

extern "C" ah_status ah_rank(ah_context* ctx, const ah_array_view* v, int32_t descending, int32_t nulls_first, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  out->type = AH_UINT32;
  const ah_type t = v->type;
  const int64_t n = v->length;
  const int w = ah_type_width(t);
  if (t == AH_UTF8_VIEW || t == AH_BINARY_VIEW)  // can_rank :31-44 accepts them
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "rank of %s (no device sort for view arrays)", ah_type_name(t));
  if (t == AH_FLOAT16) return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "rank of Float16 (no device comparison kernel)");
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;  // bytes_rank :90-101
  const bool ok = is_str || t == AH_BOOL || ah_type_is_integer(t) || t == AH_FLOAT32 || t == AH_FLOAT64;
  if (!ok) return ah_fail(ctx, AH_COMPUTE_ERROR, "%s not supported in rank", ah_type_name(t));  // :68
  if (n == 0) return AH_OK;
  if (n > (int64_t)UINT32_MAX)  // `values.len().try_into().unwrap()` :77
    return ah_fail(ctx, AH_PANIC, "called `Result::unwrap()` on an `Err` value: TryFromIntError(())");
  struct DeferredOff {  // the composition reads counts back between steps: run it synchronously
    ah_context* c;
    bool was;
    explicit DeferredOff(ah_context* cc) : c(cc), was(cc->deferred) { c->deferred = false; }
    ~DeferredOff() { c->deferred = was; }
  } sync_scope(ctx);
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, v, &nulls));
  if (!v->validity) nulls = 0;
  const int64_t m = n - nulls;
  const int64_t valid_start = nulls_first ? nulls : 0, null_start = nulls_first ? 0 : m;
  const uint32_t null_rank = (uint32_t)(nulls_first ? nulls : n);

  Temp perm(ctx);
  AH_TRY(ah_sort_to_indices(ctx, v, descending, nulls_first, -1, &perm.out));
  const uint32_t* pp = (const uint32_t*)perm.out.values;
  void* ranks = nullptr;
  const size_t rbytes = (size_t)n * 4;
  AH_TRY(ah_out_alloc(ctx, rbytes, &ranks));
  auto fail_free = [&](ah_status st) {
    ah_out_free(ctx, ranks, rbytes);
    return st;
  };
  auto grid_for = [](int64_t rows) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(256 * 16, ah_ceil_div(rows, 256))); };
  ah_prof_scope ps(ctx, "rank");
  if (m > 0) {
    ah_array_view idx{};
    idx.type = AH_UINT32;
    idx.length = m;
    idx.null_count = 0;
    idx.values = pp + valid_start;
    struct Scratch {
      ah_context* c;
      void* p = nullptr;
      ~Scratch() { ah_pool_free(c, p); }
    } iota{ctx};
    Temp sorted(ctx), edge(ctx), ends(ctx);
    ah_status st = ah_take(ctx, v, &idx, 0, &sorted.out);
    if (st != AH_OK) return fail_free(st);
    int64_t n_ends = 0;
    if (m > 1) {
      ah_array_view a = view_of(sorted.out), b = a;
      a.length = b.length = m - 1;
      if (is_str) b.offsets = (const uint8_t*)a.offsets + (t == AH_UTF8 ? 4 : 8);
      else if (w == 0) b.values_bit_offset = a.values_bit_offset + 1;
      else b.values = (const uint8_t*)a.values + w;
      st = ah_compare(ctx, AH_NEQ, &a, 0, &b, 0, &edge.out);
      if (st != AH_OK) return fail_free(st);
      st = ah_pool_alloc(ctx, (size_t)(m - 1) * 4, &iota.p);  // lives until the fill kernel is done: an all-true
      if (st != AH_OK) return fail_free(st);                   // `edge` makes filter return the iota itself (zero copy)
      st = ah_gen_iota_u32(ctx, (uint32_t*)iota.p, m - 1, 0);
      if (st == AH_OK) {
        ah_array_view iv{};
        iv.type = AH_UINT32;
        iv.length = m - 1;
        iv.null_count = 0;
        iv.values = iota.p;
        ah_array_view ev = view_of(edge.out);
        st = ah_filter(ctx, &iv, &ev, &ends.out);
      }
      if (st != AH_OK) return fail_free(st);
      n_ends = ends.out.length;
    }
    rank_fill_kernel<<<grid_for(m), 256, 0, ctx->stream>>>(pp + valid_start, m, (const uint32_t*)ends.out.values, n_ends,
                                                         (uint32_t)valid_start, (uint32_t*)ranks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = ah_stream_wait(ctx);  // the temporaries are released at the end of this scope
    if (e != hipSuccess) return fail_free(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in rank", hipGetErrorString(e)));
  }
  if (nulls > 0) {
    rank_const_kernel<<<grid_for(nulls), 256, 0, ctx->stream>>>(pp + null_start, nulls, null_rank, (uint32_t*)ranks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) return fail_free(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in rank", hipGetErrorString(e)));
  }
  out->length = n;
  out->values = ranks;
  out->values_bytes = (int64_t)rbytes;
  return AH_OK;
}
