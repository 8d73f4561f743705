// cmp.hip — arrow_ord::cmp on MI355X.
//
// Reference path: eq/neq/lt/lt_eq/gt/gt_eq/distinct/not_distinct
// (arrow-ord/src/cmp.rs:79-202) -> compare_op :220-382 -> apply :438-502
// (lt_eq = !lt swapped, gt = lt swapped, gt_eq = !lt, neq = !eq) -> apply_op
// :619-648 -> collect_bool :580-611 (64 compares -> one u64, optional negation).
// Floats compare in IEEE totalOrder, equality is bit equality
// (arrow-array/src/arithmetic.rs:400-410).
//
// MI355X design: every lane loads 16 bytes of each operand (V rows); the V
// compare bits of all 64 lanes are collected with V __ballot's (SGPR pairs) and
// bit-interleaved on the scalar unit into the V output words of the wave's
// 64*V-row group — 16-byte loads AND whole-word stores without touching LDS.
// Null handling follows compare_op's four cases with word-parallel bitmap ops.
#include "common.hpp"

#include <type_traits>

namespace {

enum { B_EQ = 0, B_LT = 1 };

template <typename T> struct Key { using type = T; };
template <> struct Key<double> { using type = int64_t; };
template <> struct Key<float> { using type = int32_t; };

// totalOrder key: x ^ (((x >> 63) as u64) >> 1)  (f64::total_cmp)
__device__ __forceinline__ int64_t order_key(double v) {
  int64_t b = __double_as_longlong(v);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}
__device__ __forceinline__ int32_t order_key(float v) {
  int32_t b = __float_as_int(v);
  return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
}
// f16::total_cmp / to_bits equality (half 2.7.1): the f32 rule on the 16-bit pattern
__device__ __forceinline__ int16_t order_key(ah_f16 v) {
  const int16_t b = (int16_t)v.bits;
  return (int16_t)(b ^ (int16_t)((uint16_t)(b >> 15) >> 1));
}
__device__ __forceinline__ uint16_t bits_key(ah_f16 v) { return v.bits; }
template <typename T> __device__ __forceinline__ T order_key(T v) { return v; }
__device__ __forceinline__ int64_t bits_key(double v) { return __double_as_longlong(v); }
__device__ __forceinline__ int32_t bits_key(float v) { return __float_as_int(v); }
template <typename T> __device__ __forceinline__ T bits_key(T v) { return v; }

// place bit p of the low 64 / V bits of x at bit p * V — on the VECTOR unit, per lane: lane k < V assembles output word k
// of its wave's group from the V ballots.  (The first form built all V words on the scalar unit, ~15 scalar
// instructions per spread: one scalar unit serves the four SIMDs of a CU, and a compare against a SCALAR — half the
// bytes per row of an array-array compare — ran out of scalar issue slots at 4.9 TB/s.)
template <typename T, int V> struct alignas(sizeof(T) * V) VecT { T e[V]; };

struct CmpArgs {
  const void* l;
  const void* r;
  int64_t len;
  int l_scalar, r_scalar;
  int base;  // B_EQ / B_LT on (l, r) as given (caller already swapped)
  int neg;
  unsigned long long* out;
  // small calls (<= 2048 blocks) whose result validity is `va [& vb]`: the kernel also writes those words and accumulates
  // their popcount (arith.hip does the same): compare + read-back kernel instead of compare + bitmap kernel + read-back
  int post;
  BitView va, vb;
  unsigned long long* vout;
  unsigned long long* total;
};

constexpr int CMP_G = 4;  // 64*V-row groups per wave: 2*CMP_G 16-byte loads in flight per lane

template <typename T, int V>
__global__ void __launch_bounds__(256) compare_kernel(CmpArgs a) {
  using VT = VecT<T, V>;
  const int lane = threadIdx.x & 63;
  const T* lp = (const T*)a.l;
  const T* rp = (const T*)a.r;
  T ls = a.l_scalar ? lp[0] : T{};
  T rs = a.r_scalar ? rp[0] : T{};
  const int64_t ngroups = (a.len + 64 * V - 1) / (64 * V);  // one group = 64*V rows = V output words
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t nwords = (a.len + 63) >> 6;
  __shared__ uint64_t s_b[4][CMP_G * V];  // per wave: the ballots of its CMP_G groups
  const int wave = threadIdx.x >> 6;
  for (int64_t g0 = wave0 * CMP_G; g0 < ngroups; g0 += nwaves * CMP_G) {
    VT lv[CMP_G], rv[CMP_G];
#pragma unroll
    for (int gi = 0; gi < CMP_G; ++gi) {  // all loads of the wave's CMP_G groups go out first
      int64_t i = ((g0 + gi) * 64 + lane) * V;
      if (i + V <= a.len) {
        if (!a.l_scalar) lv[gi] = ah_ld_stream<ah_nt_l(false)>((const VT*)(lp + i));
        if (!a.r_scalar) rv[gi] = ah_ld_stream<ah_nt_l(false)>((const VT*)(rp + i));
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          lv[gi].e[e] = (!a.l_scalar && i + e < a.len) ? lp[i + e] : T{};
          rv[gi].e[e] = (!a.r_scalar && i + e < a.len) ? rp[i + e] : T{};
        }
      }
    }
#pragma unroll
    for (int gi = 0; gi < CMP_G; ++gi) {
      const int64_t i = ((g0 + gi) * 64 + lane) * V;  // (groups past the end compare nothing: every row test fails)
#pragma unroll
      for (int e = 0; e < V; ++e) {
        T x = a.l_scalar ? ls : lv[gi].e[e];
        T y = a.r_scalar ? rs : rv[gi].e[e];
        bool res = a.base == B_EQ ? (bits_key(x) == bits_key(y)) : (order_key(x) < order_key(y));
        res = res && (i + e < a.len);
        const uint64_t b = __ballot(res);
        if (lane == 0) s_b[wave][gi * V + e] = b;
      }
    }
    // The CMP_G * V output words of the wave's groups, one per lane: word j = group j / V, lanes [k * 64 / V, (k + 1) * 64 / V)
    // with k = j % V — lane j interleaves its 64 / V-bit slices of that group's V ballots.  (Same-wave LDS traffic: the
    // writes above are ordered before these reads by the wave's own program order.)
    if (lane < CMP_G * V) {
      const int gi = lane / V, ksh = (lane % V) * (64 / V);
      uint64_t mine = 0;
#pragma unroll
      for (int e = 0; e < V; ++e) mine |= vspread<V>(s_b[wave][gi * V + e] >> ksh) << e;
      if (a.neg) mine = ~mine;  // collect_bool negates whole words, padding included (cmp.rs:590-592)
      const int64_t wi = g0 * V + lane;
      if (wi < nwords) a.out[wi] = mine;
    }
  }
  if (a.post) {  // uniform
    unsigned long long acc = 0;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
      uint64_t r = bv_fetch64(a.va, w << 6, a.len);
      if (a.vb.words) r &= bv_fetch64(a.vb, w << 6, a.len);
      a.vout[w] = r;
      acc += __popcll(r);
    }
    acc = wave_reduce_add64(acc);
    __shared__ unsigned long long sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ah_count_add(a.total, sm[0] + sm[1] + sm[2] + sm[3]);
  }
}

template <typename T>
void launch_cmp_t(ah_context* ctx, const CmpArgs& a, bool aligned) {
  constexpr int VV = sizeof(T) >= 4 ? 16 / sizeof(T) : 4;  // 1/2-byte types: 4 rows per lane
  const int V = aligned ? VV : 1;
  int64_t ngroups = ah_ceil_div(a.len, 64 * (int64_t)V);
  int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(ngroups, 4 * CMP_G), (int64_t)1 << 30));
  if (aligned) compare_kernel<T, VV><<<grid, 256, 0, ctx->stream>>>(a);
  else compare_kernel<T, 1><<<grid, 256, 0, ctx->stream>>>(a);
}

// Boolean operands: word-parallel (ArrayOrd for &BooleanArray, cmp.rs:740-761: lt = !l & r)
__global__ void __launch_bounds__(256) compare_bool_kernel(BitView l, BitView r, int64_t len,
                                                           int l_scalar, int r_scalar, int base, int neg,
                                                           unsigned long long* out) {
  int64_t nwords = (len + 63) >> 6;
  uint64_t ls = l_scalar ? (bv_get(l, 0) ? ~0ull : 0ull) : 0ull;
  uint64_t rs = r_scalar ? (bv_get(r, 0) ? ~0ull : 0ull) : 0ull;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
    int64_t s = w << 6;
    uint64_t x = l_scalar ? ls : bv_fetch64(l, s, len);
    uint64_t y = r_scalar ? rs : bv_fetch64(r, s, len);
    uint64_t res = base == B_EQ ? ~(x ^ y) : (~x & y);
    int64_t rem = len - s;
    if (rem < 64) res &= (1ull << rem) - 1;
    out[w] = neg ? ~res : res;
  }
}

const char* cmp_sym(int op) {  // Display for Op (cmp.rs:55-68)
  switch (op) {
    case AH_EQ: return "==";
    case AH_NEQ: return "!=";
    case AH_LT: return "<";
    case AH_LT_EQ: return "<=";
    case AH_GT: return ">";
    case AH_GT_EQ: return ">=";
    case AH_DISTINCT: return "IS DISTINCT FROM";
    default: return "IS NOT DISTINCT FROM";
  }
}

}  // namespace

namespace {

// ArrayOrd for &GenericByteArray (arrow-ord/src/cmp.rs:783-830): eq = same length and bytes, lt = bytewise
// lexicographic.  Lane per row; 8 bytes at a time (big-endian compare of unaligned words), byte tail; the result
// word of a wave's 64 rows is a ballot.
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

template <typename O>
__global__ __launch_bounds__(256) void compare_bytes_kernel(const O* lo, const uint8_t* ld, const O* ro, const uint8_t* rd,
                                                           int64_t len, int l_scalar, int r_scalar, int base, int neg,
                                                           unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (len + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t row = w * 64 + lane;
    bool res = false;
    if (row < len) {
      const int64_t li = l_scalar ? 0 : row, ri = r_scalar ? 0 : row;
      const int64_t a0 = (int64_t)lo[li], b0 = (int64_t)ro[ri];
      const int64_t na = (int64_t)lo[li + 1] - a0, nb = (int64_t)ro[ri + 1] - b0;
      const uint8_t* pa = ld + a0;
      const uint8_t* pb = rd + b0;
      if (base == B_EQ && na != nb) {
        res = false;
      } else {
        const int64_t n = na < nb ? na : nb;
        int64_t i = 0;
        int c = 0;  // sign of the first differing position
        for (; i + 8 <= n; i += 8) {
          const uint64_t x = load_u64_unaligned(pa + i), y = load_u64_unaligned(pb + i);
          if (x != y) {
            const uint64_t bx = __builtin_bswap64(x), by = __builtin_bswap64(y);
            c = bx < by ? -1 : 1;
            break;
          }
        }
        if (c == 0)
          for (; i < n; ++i) {
            const int d = (int)pa[i] - (int)pb[i];
            if (d) {
              c = d < 0 ? -1 : 1;
              break;
            }
          }
        res = base == B_EQ ? (c == 0) : (c < 0 || (c == 0 && na < nb));
      }
      res = res != (neg != 0);
    } else {
      res = neg != 0;  // collect_bool: negated padding bits are ones (cmp.rs:600-607)
    }
    const unsigned long long word = __ballot(res);
    if (lane == 0) out[w] = word;
  }
}

}  // namespace

extern "C" ah_status ah_compare(ah_context* ctx, ah_cmp_op op, const ah_array_view* lhs, int32_t l_s,
                                const ah_array_view* rhs, int32_t r_s, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !lhs || !rhs || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_EQ || op > AH_NOT_DISTINCT) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown comparison op %d", op);
  l_s = l_s != 0;
  r_s = r_s != 0;
  // compare_op (cmp.rs:228-264)
  if (lhs->length != rhs->length && !l_s && !r_s)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Cannot compare arrays of different lengths, got %lld vs %lld",
                   (long long)lhs->length, (long long)rhs->length);
  const int64_t len = l_s ? rhs->length : lhs->length;
  if (lhs->type != rhs->type)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid comparison operation: %s %s %s",
                   ah_type_name(lhs->type), cmp_sym(op), ah_type_name(rhs->type));
  const ah_type t = lhs->type;
  const bool is_bytes = t == AH_UTF8 || t == AH_LARGE_UTF8;
  // AH_FIXED16 compares as i128 (Decimal128; the host checks that precision and scale agree, compare_op cmp.rs:243-249).
  // IntervalMonthDayNano shares the layout but orders by (months, days, nanoseconds): its host must not come here.
  if (!(t == AH_BOOL || ah_type_is_integer(t) || ah_type_is_float(t) || t == AH_FLOAT16 || is_bytes || t == AH_FIXED16))
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "comparison not supported for type %s", ah_type_name(t));
  if ((l_s && lhs->length < 1) || (r_s && rhs->length < 1))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar datum must have length 1");
  out->type = AH_BOOL;
  out->length = len;
  if (len == 0) return AH_OK;

  // .filter(null_count > 0).  Deferred mode: an array side whose null count is still unknown counts as
  // nullable instead of being counted (a host sync); the result bits are the same, the validity is kept.
  auto nullable = [&](const ah_array_view* v, bool is_scalar, bool* yes) -> ah_status {
    *yes = false;
    if (!v->validity) return AH_OK;
    if (ctx->deferred && !is_scalar && v->null_count < 0) {
      *yes = true;
      return AH_OK;
    }
    int64_t n = 0;
    AH_TRY(ah_resolve_null_count(ctx, v, &n));
    *yes = n > 0;
    return AH_OK;
  };
  bool lnul = false, rnul = false;
  AH_TRY(nullable(lhs, l_s, &lnul));
  AH_TRY(nullable(rhs, r_s, &rnul));
  const bool distinct_op = op == AH_DISTINCT || op == AH_NOT_DISTINCT;

  const size_t bytes = ah_bitmap_bytes(len);
  unsigned long long* vals = nullptr;   // comparison bits
  unsigned long long* nb = nullptr;     // result validity
  unsigned long long* tmp = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, (void**)&vals));
  auto fail_free = [&](ah_status st) {
    ah_out_free(ctx, vals, bytes);
    ah_out_free(ctx, nb, bytes);
    ah_pool_free(ctx, tmp);
    return st;
  };

  // which cases need the value comparison at all (the reference defers it: cmp.rs:300)
  const bool scalar_null = (lnul && l_s && !(rnul && r_s)) || (rnul && r_s && !(lnul && l_s));
  const bool need_values = !scalar_null || (lnul && rnul && l_s == r_s);

  // fused validity for small primitive calls: set by the two arms below that would otherwise launch a counted bitmap op
  BitView f_va{nullptr, 0}, f_vb{nullptr, 0};
  unsigned long long* f_vout = nullptr;
  bool fused_done = false;
  auto run_values = [&](unsigned long long* dst) -> ah_status {
    // apply (cmp.rs:480-488)
    int base, neg;
    bool swap;
    switch (op) {
      case AH_EQ: case AH_NOT_DISTINCT: base = B_EQ; neg = 0; swap = false; break;
      case AH_NEQ: case AH_DISTINCT: base = B_EQ; neg = 1; swap = false; break;
      case AH_LT: base = B_LT; neg = 0; swap = false; break;
      case AH_LT_EQ: base = B_LT; neg = 1; swap = true; break;
      case AH_GT: base = B_LT; neg = 0; swap = true; break;
      default: base = B_LT; neg = 1; swap = false; break;  // GT_EQ
    }
    const ah_array_view* L = swap ? rhs : lhs;
    const ah_array_view* R = swap ? lhs : rhs;
    int Ls = swap ? r_s : l_s, Rs = swap ? l_s : r_s;
    ah_prof_scope ps(ctx, "compare");
    if (is_bytes) {
      const int64_t nwords = (len + 63) >> 6;
      const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(256 * 16, ah_ceil_div(nwords, 4)));
      if (t == AH_UTF8)
        compare_bytes_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t*)L->offsets, (const uint8_t*)L->values,
                                                                     (const int32_t*)R->offsets, (const uint8_t*)R->values, len,
                                                                     Ls, Rs, base, neg, dst);
      else
        compare_bytes_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t*)L->offsets, (const uint8_t*)L->values,
                                                                     (const int64_t*)R->offsets, (const uint8_t*)R->values, len,
                                                                     Ls, Rs, base, neg, dst);
      return AH_OK;
    }
    if (t == AH_BOOL) {
      int64_t nwords = (len + 63) >> 6;
      int grid = (int)std::min<int64_t>(4096, ah_ceil_div(nwords, 256));
      compare_bool_kernel<<<grid, 256, 0, ctx->stream>>>(
          make_bitview(L->values, L->values_bit_offset), make_bitview(R->values, R->values_bit_offset),
          len, Ls, Rs, base, neg, dst);
      return AH_OK;
    }
    CmpArgs a{};
    a.l = L->values;
    a.r = R->values;
    a.len = len;
    a.l_scalar = Ls;
    a.r_scalar = Rs;
    a.base = base;
    a.neg = neg;
    a.out = dst;
    bool aligned = true;
    if (!Ls) aligned = aligned && (((uintptr_t)a.l & 15) == 0);
    if (!Rs) aligned = aligned && (((uintptr_t)a.r & 15) == 0);
    if (f_vout && !ctx->deferred) {
      const int w8 = ah_type_width(t);
      const int vv = aligned ? (w8 >= 4 ? 16 / w8 : 4) : 1;
      if (ah_ceil_div(ah_ceil_div(len, 64 * (int64_t)vv), 4 * CMP_G) <= 2048) {
        a.post = 1;
        a.va = f_va, a.vb = f_vb, a.vout = f_vout;
        a.total = ctx->scratch + AH_TICKET_COUNT;
        fused_done = true;
      }
    }
    switch (t) {
      case AH_INT8: launch_cmp_t<int8_t>(ctx, a, aligned); break;
      case AH_INT16: launch_cmp_t<int16_t>(ctx, a, aligned); break;
      case AH_INT32: launch_cmp_t<int32_t>(ctx, a, aligned); break;
      case AH_INT64: launch_cmp_t<int64_t>(ctx, a, aligned); break;
      case AH_UINT8: launch_cmp_t<uint8_t>(ctx, a, aligned); break;
      case AH_UINT16: launch_cmp_t<uint16_t>(ctx, a, aligned); break;
      case AH_UINT32: launch_cmp_t<uint32_t>(ctx, a, aligned); break;
      case AH_UINT64: launch_cmp_t<uint64_t>(ctx, a, aligned); break;
      case AH_FLOAT16: launch_cmp_t<ah_f16>(ctx, a, aligned); break;
      case AH_FLOAT32: launch_cmp_t<float>(ctx, a, aligned); break;
      case AH_FIXED16: launch_cmp_t<__int128>(ctx, a, aligned); break;
      default: launch_cmp_t<double>(ctx, a, aligned); break;
    }
    return AH_OK;
  };

  BitView lv = make_bitview(lhs->validity, lhs->validity_bit_offset);
  BitView rv = make_bitview(rhs->validity, rhs->validity_bit_offset);
  const BitView none{nullptr, 0};
  int64_t set_bits = len;
  bool count_known = !ctx->deferred;  // deferred: only the all-null results know their count
  bool has_nb = false, waited = false;
  ah_status st = AH_OK;

  if (lnul && rnul && l_s == r_s) {
    // both nullable, both or neither scalar (cmp.rs:322-347)
    if (distinct_op) {
      st = ah_pool_alloc(ctx, bytes, (void**)&tmp);
      if (st != AH_OK) return fail_free(st);
      st = run_values(tmp);
      BitView tv = make_bitview(tmp, 0);
      if (st == AH_OK)
        st = ah_bitmap_op(ctx, op == AH_DISTINCT ? BM_DISTINCT_BOTH : BM_NOT_DISTINCT_BOTH, lv, rv, tv, len,
                          vals, nullptr);
    } else {
      st = ah_out_alloc(ctx, bytes, (void**)&nb);
      if (st == AH_OK) {
        f_va = lv, f_vb = rv, f_vout = nb;
        st = run_values(vals);
      }
      if (st == AH_OK && fused_done) {
        hipError_t fe = ah_count_read(ctx, &set_bits);
        if (fe != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "compare kernel failed: %s", hipGetErrorString(fe));
      } else if (fused_done) {
        ah_count_reset(ctx);  // enqueued with its counting tail but never read: the counters must not stay dirty
      } else if (st == AH_OK) {
        st = ah_bitmap_op(ctx, BM_AND, lv, rv, none, len, nb, AH_COUNT(ctx, &set_bits));
      }
      waited = st == AH_OK && !ctx->deferred;  // the popcount read-back waited for the stream: it ends the call
      has_nb = true;
    }
  } else if (lnul && rnul) {
    // scalar is null, other side non-scalar and nullable (cmp.rs:349-356)
    BitView av = l_s ? rv : lv;
    if (op == AH_DISTINCT) st = ah_bitmap_op(ctx, BM_COPY, av, none, none, len, vals, nullptr);
    else if (op == AH_NOT_DISTINCT) st = ah_bitmap_op(ctx, BM_NOT, av, none, none, len, vals, nullptr);
    else {  // BooleanArray::new_null(len)
      st = ah_out_alloc(ctx, bytes, (void**)&nb);
      if (st == AH_OK) {
        hipMemsetAsync(vals, 0, bytes, ctx->stream);
        hipMemsetAsync(nb, 0, bytes, ctx->stream);
        set_bits = 0;
        count_known = true;
        has_nb = true;
      }
    }
  } else if (lnul || rnul) {
    // only one side nullable (cmp.rs:357-378)
    const bool is_scalar = lnul ? l_s : r_s;
    BitView nv = lnul ? lv : rv;
    if (is_scalar) {
      if (op == AH_DISTINCT) {  // BooleanBuffer::new_set(len)
        st = ah_bitmap_op(ctx, BM_COPY, none, none, none, len, vals, nullptr);
      } else if (op == AH_NOT_DISTINCT) {
        hipMemsetAsync(vals, 0, bytes, ctx->stream);
      } else {
        st = ah_out_alloc(ctx, bytes, (void**)&nb);
        if (st == AH_OK) {
          hipMemsetAsync(vals, 0, bytes, ctx->stream);
          hipMemsetAsync(nb, 0, bytes, ctx->stream);
          set_bits = 0;
          count_known = true;
          has_nb = true;
        }
      }
    } else if (distinct_op) {
      st = ah_pool_alloc(ctx, bytes, (void**)&tmp);
      if (st != AH_OK) return fail_free(st);
      st = run_values(tmp);
      BitView tv = make_bitview(tmp, 0);
      if (st == AH_OK)
        st = ah_bitmap_op(ctx, op == AH_DISTINCT ? BM_ORNOT : BM_AND, nv, tv, none, len, vals, nullptr);
    } else {
      st = ah_out_alloc(ctx, bytes, (void**)&nb);
      if (st == AH_OK) {
        f_va = nv, f_vb = none, f_vout = nb;
        st = run_values(vals);
      }
      if (st == AH_OK && fused_done) {
        hipError_t fe = ah_count_read(ctx, &set_bits);
        if (fe != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "compare kernel failed: %s", hipGetErrorString(fe));
      } else if (fused_done) {
        ah_count_reset(ctx);  // enqueued with its counting tail but never read: the counters must not stay dirty
      } else if (st == AH_OK) {
        st = ah_bitmap_op(ctx, BM_COPY, nv, none, none, len, nb, AH_COUNT(ctx, &set_bits));
      }
      waited = st == AH_OK && !ctx->deferred;
      has_nb = true;
    }
  } else {
    st = run_values(vals);  // neither side nullable (:380)
  }
  (void)need_values;
  if (st != AH_OK) return fail_free(st);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !waited) e = ah_end_of_call_sync(ctx);
  if (e != hipSuccess) {
    fail_free(AH_HIP_ERROR);
    return ah_fail(ctx, AH_HIP_ERROR, "compare kernel failed: %s", hipGetErrorString(e));
  }
  ah_pool_free(ctx, tmp);
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (has_nb) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bytes;
    out->null_count = count_known ? len - set_bits : -1;
  }
  return AH_OK;
}
