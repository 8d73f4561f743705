// arrow_arith::aggregate on MI355X: sum / product / min / max / bit_and / bit_or / bit_xor over
// null-bitmapped PrimitiveArrays, sum_checked / product_checked, min_boolean / max_boolean.
//
// Reference: arrow-arith/src/aggregate.rs — accumulators :51-177, lane kernels :179-297, dispatch
// `aggregate` :317-361, `sum` :943, `product` :953, `min` :1012, `max` :1027, `sum_checked` :897,
// `product_checked` :963, `bit_and/or/xor` :776-873, `min_boolean/max_boolean` :372-457.
//
// Roofline: one streaming read of the values (+ 1 bit/row of validity) and an 8-byte result — HBM
// read bound.  Two kernels per call: `agg_kernel` (grid-stride 16-byte loads, per-thread accumulator,
// wave shuffle tree, one partial per workgroup) and a one-workgroup `agg_final`.
//
//  * Integer sum/product wrap (add_wrapping / mul_wrapping), so any association order gives the
//    reference's bits; narrow types accumulate in 32 bits and truncate (same residue class).
//  * min/max follow `is_lt` = total order for floats (arithmetic.rs:400): every element is mapped to
//    an unsigned key whose integer order IS that order (sign flip for signed ints; for floats
//    negative -> ~bits, positive -> bits ^ signbit), reduced with unsigned min/max and mapped back.
//    The accumulator identities are the keys of MAX_TOTAL_ORDER / MIN_TOTAL_ORDER (:117,:151).
//  * Float sum/product: the reference's own bits depend on its compile-time lane count
//    (PREFERRED_VECTOR_SIZE :300-307), so it defines no single answer; this kernel is a fixed tree
//    (deterministic for a given length and alignment) inside the usual pairwise-summation bound.
//  * sum_checked / product_checked are *sequential* in the reference: they fail as soon as a PREFIX
//    leaves the type's range, and the message quotes that prefix and the offending element.  Both are
//    evaluated exactly with ORDERED reductions over associative (non-commutative) summaries:
//      sum:     (s, max prefix, min prefix) in 128-bit arithmetic;
//      product: |q| of the elements before the first zero (saturating), sign, "a maximal-magnitude
//               prefix is positive / negative", has-zero — enough because |prefix| is nondecreasing
//               up to the first zero and 0 absorbs afterwards; only +2^(w-1) vs -2^(w-1) needs signs.
//    A one-thread pass over the <= 2048 workgroup summaries finds the first failing tile, and a
//    one-thread kernel re-walks only that tile to recover the exact operands of the error message.
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.hpp"

namespace {

enum { K_SUM = 0, K_PROD = 1, K_MIN = 2, K_MAX = 3, K_AND = 4, K_OR = 5, K_XOR = 6 };
enum { M_UNSIGNED = 0, M_SIGNED = 1, M_FLOAT = 2 };

constexpr int AGG_BLOCK = 256;
constexpr int AGG_MAX_GRID = 2048;

template <int W> struct UOf;
template <> struct UOf<1> { using type = uint32_t; };
template <> struct UOf<2> { using type = uint32_t; };
template <> struct UOf<4> { using type = uint32_t; };
template <> struct UOf<8> { using type = uint64_t; };

// element e of a 16-byte vector, zero-extended
template <int W> __device__ __forceinline__ typename UOf<W>::type vec_elem(const uint4& q, int e) {
  const uint32_t d[4] = {q.x, q.y, q.z, q.w};
  if constexpr (W == 8) return (uint64_t)d[2 * e] | ((uint64_t)d[2 * e + 1] << 32);
  else if constexpr (W == 4) return d[e];
  else if constexpr (W == 2) return (d[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
  else return (d[e >> 2] >> ((e & 3) * 8)) & 0xFFu;
}

// total-order key <-> raw bits (W-byte wide values held zero-extended)
template <int W, int M, typename U> __device__ __host__ __forceinline__ U to_key(U b) {
  constexpr U sign = (U)1 << (W * 8 - 1);
  constexpr U mask = (W == 8) ? ~(U)0 : (((U)1 << (W * 8 - 1)) << 1) - 1;
  if (M == M_UNSIGNED) return b;
  if (M == M_SIGNED) return b ^ sign;
  return (b & sign) ? (~b & mask) : (b ^ sign);
}
template <int W, int M, typename U> __device__ __host__ __forceinline__ U from_key(U k) {
  constexpr U sign = (U)1 << (W * 8 - 1);
  constexpr U mask = (W == 8) ? ~(U)0 : (((U)1 << (W * 8 - 1)) << 1) - 1;
  if (M == M_UNSIGNED) return k;
  if (M == M_SIGNED) return k ^ sign;
  return (k & sign) ? (k ^ sign) : (~k & mask);
}

template <typename A, int K> __device__ __host__ __forceinline__ A agg_ident() {
  if (K == K_SUM || K == K_OR || K == K_XOR || K == K_MAX) return (A)0;
  if (K == K_PROD) return (A)1;
  if constexpr (std::is_floating_point<A>::value) return (A)0;
  else return (A) ~(A)0;  // K_MIN, K_AND
}
template <typename A, int K> __device__ __host__ __forceinline__ A agg_combine(A a, A b) {
  if constexpr (std::is_floating_point<A>::value) {
    return K == K_SUM ? a + b : a * b;
  } else {
    switch (K) {
      case K_SUM: return a + b;
      case K_PROD: return a * b;
      case K_MIN: return b < a ? b : a;
      case K_MAX: return b > a ? b : a;
      case K_AND: return a & b;
      case K_OR: return a | b;
      default: return a ^ b;
    }
  }
}

// raw zero-extended element -> accumulator domain
template <int W, typename A, int K, int M>
__device__ __forceinline__ A agg_lift(typename UOf<W>::type raw) {
  if constexpr (std::is_same<A, float>::value) return __uint_as_float((uint32_t)raw);
  else if constexpr (std::is_same<A, double>::value) return __longlong_as_double((long long)raw);
  else if constexpr (K == K_MIN || K == K_MAX) return (A)to_key<W, M, typename UOf<W>::type>(raw);
  else return (A)raw;
}

template <typename A> __device__ __forceinline__ A shfl_xor_any(A v, int o) {
  if constexpr (sizeof(A) == 8) {
    long long x;
    memcpy(&x, &v, 8);
    x = __shfl_xor(x, o, 64);
    memcpy(&v, &x, 8);
    return v;
  } else {
    int x;
    memcpy(&x, &v, 4);
    x = __shfl_xor(x, o, 64);
    memcpy(&v, &x, 4);
    return v;
  }
}

// The array is addressed in a 16-byte aligned frame: vector g holds rows [g*E - skip, g*E - skip + E).
// A workgroup streams whole 32 KiB tiles (8 x 16-byte loads in flight per thread).  The validity words
// of a tile are requested together with its values — loading them one by one inside the accumulate
// loop serialised eight L2 round trips per tile and cost 22 % (1.57 -> 1.29 ms on 1e9 Int64 rows).
template <int W, typename A, int K, int M, bool HAS_VALID>
__global__ __launch_bounds__(AGG_BLOCK) void agg_kernel(const uint4* __restrict__ base, int64_t skip, int64_t len,
                                                        BitView valid, int64_t nvec, int may_straddle,
                                                        A* __restrict__ partials) {
  constexpr int E = 16 / W;
  constexpr int U = 8;
  constexpr uint32_t FULL = (1u << E) - 1u;
  A acc = agg_ident<A, K>();
  auto accumulate = [&](const uint4& q, uint32_t bits) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const A x = agg_lift<W, A, K, M>(vec_elem<W>(q, e));
      acc = agg_combine<A, K>(acc, ((bits >> e) & 1u) ? x : agg_ident<A, K>());
    }
  };
  auto edge = [&](int64_t g) {  // vectors that may hang over either end of the array
    const int64_t r0 = g * E - skip;
    uint32_t bits = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t r = r0 + e;
      if (r >= 0 && r < len && (!HAS_VALID || bv_get(valid, r))) bits |= 1u << e;
    }
    accumulate(base[g], bits);
  };
  constexpr int64_t TILE_V = (int64_t)U * AGG_BLOCK;
  const int64_t full_tiles = nvec / TILE_V;
  const int64_t last_word = HAS_VALID ? ((valid.off + len - 1) >> 6) : 0;
  for (int64_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
    const int64_t g0 = tile * TILE_V + threadIdx.x;
    const bool interior = tile * TILE_V * E - skip >= 0 && (tile + 1) * TILE_V * E - skip <= len;
    if (!interior) {
      for (int u = 0; u < U; ++u) edge(g0 + u * AGG_BLOCK);
      continue;
    }
    uint4 q[U];
    uint64_t lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      q[u] = base[g0 + u * AGG_BLOCK];
      if (HAS_VALID) {
        const int64_t wi = (valid.off + (g0 + u * AGG_BLOCK) * E - skip) >> 6;
        lo[u] = valid.words[wi];
        if (may_straddle) hi[u] = valid.words[wi < last_word ? wi + 1 : last_word];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t bits = FULL;
      if (HAS_VALID) {
        const int sh = (int)((valid.off + (g0 + u * AGG_BLOCK) * E - skip) & 63);
        uint64_t b = lo[u] >> sh;
        if (may_straddle && sh) b |= hi[u] << (64 - sh);
        bits = (uint32_t)b & FULL;
      }
      accumulate(q[u], bits);
    }
  }
  if (blockIdx.x == (unsigned)(full_tiles % gridDim.x))  // the ragged tail, at most one tile
    for (int64_t g = full_tiles * TILE_V + threadIdx.x; g < nvec; g += AGG_BLOCK) edge(g);

#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc = agg_combine<A, K>(acc, shfl_xor_any<A>(acc, o));
  __shared__ A sm[AGG_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = sm[0];
#pragma unroll
    for (int w = 1; w < AGG_BLOCK / 64; ++w) r = agg_combine<A, K>(r, sm[w]);
    partials[blockIdx.x] = r;
  }
}

template <typename A, int K>
__global__ __launch_bounds__(AGG_BLOCK) void agg_final(const A* __restrict__ partials, int n, A* __restrict__ out) {
  A acc = agg_ident<A, K>();
  for (int i = threadIdx.x; i < n; i += AGG_BLOCK) acc = agg_combine<A, K>(acc, partials[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc = agg_combine<A, K>(acc, shfl_xor_any<A>(acc, o));
  __shared__ A sm[AGG_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = sm[0];
#pragma unroll
    for (int w = 1; w < AGG_BLOCK / 64; ++w) r = agg_combine<A, K>(r, sm[w]);
    out[0] = r;
  }
}

struct Frame {
  const uint4* base;
  int64_t skip, nvec;
  int grid;
};
Frame make_frame(const void* values, int w, int64_t len) {
  Frame f;
  const uintptr_t p = (uintptr_t)values;
  const uintptr_t al = p & ~(uintptr_t)15;
  f.base = (const uint4*)al;
  f.skip = (int64_t)((p - al) / w);
  const int e = 16 / w;
  f.nvec = (f.skip + len + e - 1) / e;
  int64_t g = (f.nvec + AGG_BLOCK * 8 - 1) / (AGG_BLOCK * 8);
  f.grid = (int)std::max<int64_t>(1, std::min<int64_t>(g, AGG_MAX_GRID));
  return f;
}

template <int W, typename A, int K, int M>
ah_status run_agg(ah_context* ctx, const ah_array_view* v, bool has_valid, void* scratch, uint64_t* raw_out) {
  const Frame f = make_frame(v->values, W, v->length);
  A* partials = (A*)scratch;
  A* result = partials + AGG_MAX_GRID;
  const BitView bv = make_bitview(v->validity, v->validity_bit_offset);
  // every vector's first validity bit sits at (off - skip) mod E inside its E-bit group: when that is 0
  // no group crosses a 64-bit word
  const int may_straddle = has_valid && (((bv.off - f.skip) % (16 / W)) != 0);
  {
    ah_prof_scope ps(ctx, "aggregate");
    if (has_valid)
      hipLaunchKernelGGL((agg_kernel<W, A, K, M, true>), dim3(f.grid), dim3(AGG_BLOCK), 0, ctx->stream, f.base, f.skip,
                         v->length, bv, f.nvec, may_straddle, partials);
    else
      hipLaunchKernelGGL((agg_kernel<W, A, K, M, false>), dim3(f.grid), dim3(AGG_BLOCK), 0, ctx->stream, f.base,
                         f.skip, v->length, bv, f.nvec, may_straddle, partials);
  }
  hipLaunchKernelGGL((agg_final<A, K>), dim3(1), dim3(AGG_BLOCK), 0, ctx->stream, partials, f.grid, result);
  AH_HIP(ctx, hipGetLastError());
  AH_HIP(ctx, ah_d2h_wait(ctx, ctx->pinned, result, sizeof(A)));
  A r;
  memcpy(&r, ctx->pinned, sizeof(A));
  uint64_t raw = 0;
  if constexpr (std::is_floating_point<A>::value) {
    memcpy(&raw, &r, sizeof(A));
  } else if (K == K_MIN || K == K_MAX) {
    raw = (uint64_t)from_key<W, M, typename UOf<W>::type>((typename UOf<W>::type)r);
  } else {
    raw = (uint64_t)r;
  }
  *raw_out = raw;
  return AH_OK;
}

template <int W, int M>
ah_status run_minmax(ah_context* ctx, int k, const ah_array_view* v, bool hv, void* s, uint64_t* raw) {
  using A = typename UOf<W>::type;
  return k == K_MIN ? run_agg<W, A, K_MIN, M>(ctx, v, hv, s, raw) : run_agg<W, A, K_MAX, M>(ctx, v, hv, s, raw);
}
template <int W> ah_status run_int(ah_context* ctx, int k, int m, const ah_array_view* v, bool hv, void* s, uint64_t* raw) {
  using A = typename UOf<W>::type;
  switch (k) {
    case K_SUM: return run_agg<W, A, K_SUM, M_UNSIGNED>(ctx, v, hv, s, raw);
    case K_PROD: return run_agg<W, A, K_PROD, M_UNSIGNED>(ctx, v, hv, s, raw);
    case K_AND: return run_agg<W, A, K_AND, M_UNSIGNED>(ctx, v, hv, s, raw);
    case K_OR: return run_agg<W, A, K_OR, M_UNSIGNED>(ctx, v, hv, s, raw);
    case K_XOR: return run_agg<W, A, K_XOR, M_UNSIGNED>(ctx, v, hv, s, raw);
    default:
      if (m == M_SIGNED) return run_minmax<W, M_SIGNED>(ctx, k, v, hv, s, raw);
      if (m == M_FLOAT) return run_minmax<W, M_FLOAT>(ctx, k, v, hv, s, raw);
      return run_minmax<W, M_UNSIGNED>(ctx, k, v, hv, s, raw);
  }
}

// ------------------------------------------------------------------ ordered (checked) reductions
typedef __int128 i128;
typedef unsigned __int128 u128;

struct SumS {  // summary of a run of addends: total, max and min over its non-empty prefixes
  i128 s, mx, mn;
};
struct SumMonoid {
  using S = SumS;
  static __device__ __host__ __forceinline__ S identity() {
    return S{0, -((i128)1 << 120), ((i128)1 << 120)};
  }
  static __device__ __host__ __forceinline__ S combine(const S& a, const S& b) {
    const i128 bm = a.s + b.mx, bn = a.s + b.mn;
    return S{a.s + b.s, a.mx > bm ? a.mx : bm, a.mn < bn ? a.mn : bn};
  }
  static __device__ __forceinline__ S lift(i128 x) { return S{x, x, x}; }
};

struct ProdS {  // summary of a run of factors (see the header comment)
  uint64_t mag;     // |product of the factors before the first zero|, exact unless `sat`
  uint32_t flags;   // F_*
};
enum { F_NEG = 1, F_MAXPOS = 2, F_MAXNEG = 4, F_ZERO = 8, F_SAT = 16 };
struct ProdMonoid {
  using S = ProdS;
  static __device__ __host__ __forceinline__ S identity() { return S{1, 0}; }
  static __device__ __host__ __forceinline__ S combine(const S& a, const S& b) {
    if (a.flags & F_ZERO) return a;  // everything after a zero multiplies 0: never overflows
    S r;
    const u128 m = (u128)a.mag * (u128)b.mag;
    const bool sat = (a.flags & F_SAT) || (b.flags & F_SAT) || (uint64_t)(m >> 64) != 0;
    r.mag = sat ? ~0ull : (uint64_t)m;
    const bool aneg = a.flags & F_NEG;
    uint32_t f = ((a.flags ^ b.flags) & F_NEG) | (b.flags & F_ZERO) | (sat ? F_SAT : 0);
    // prefixes of b seen through a's sign
    const bool bpos = b.flags & (aneg ? F_MAXNEG : F_MAXPOS), bneg = b.flags & (aneg ? F_MAXPOS : F_MAXNEG);
    const bool keep_a = (b.mag == 1) && !(b.flags & F_SAT);  // b never grew the magnitude
    if (bpos || (keep_a && (a.flags & F_MAXPOS))) f |= F_MAXPOS;
    if (bneg || (keep_a && (a.flags & F_MAXNEG))) f |= F_MAXNEG;
    r.flags = f;
    return r;
  }
  static __device__ __forceinline__ S lift(i128 x) {
    if (x == 0) return S{1, F_ZERO};
    if (x < 0) return S{(uint64_t)(u128)(-x), F_NEG | F_MAXNEG};
    return S{(uint64_t)(u128)x, F_MAXPOS};
  }
};

template <typename S> __device__ __forceinline__ S shfl_down_struct(const S& v, int o) {
  constexpr int N = sizeof(S) / 4;
  int w[N];
  memcpy(w, &v, sizeof(S));
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = __shfl_down(w[i], o, 64);
  S r;
  memcpy(&r, w, sizeof(S));
  return r;
}

template <int W, bool SIGNED> __device__ __forceinline__ i128 widen(typename UOf<W>::type raw) {
  if constexpr (!SIGNED) return (i128)(u128)raw;
  else if constexpr (W == 8) return (i128)(int64_t)raw;
  else if constexpr (W == 4) return (i128)(int32_t)raw;
  else if constexpr (W == 2) return (i128)(int16_t)raw;
  else return (i128)(int8_t)raw;
}

// Workgroup b owns the contiguous vectors [b*tile, (b+1)*tile); wave w of it a contiguous quarter.  One wave
// iteration covers 64*UL consecutive vectors: lane l takes the UL vectors [g0 + l*UL, +UL) (64 bytes), folds
// their rows sequentially, and a shfl_down tree that always puts the lower lane on the left finishes the
// ordered reduction.  (One vector per lane made the tree 6x more expensive than the loads: 6.7 ms / 1e9 rows.)
// Null rows are replaced by the operation's neutral ELEMENT (0 / 1): that re-records an already seen prefix
// (or the empty prefix 0 / 1, always in range), so the overflow verdict is unchanged.
constexpr int ord_ul(int w) { return 16 / w <= 8 ? 8 : 4; }  // UL * E validity bits must fit one u64

template <int W, bool SIGNED, typename MON, bool HAS_VALID>
__global__ __launch_bounds__(AGG_BLOCK) void ordered_kernel(const uint4* __restrict__ base, int64_t skip, int64_t len,
                                                            BitView valid, int64_t nvec, int64_t tile,
                                                            typename MON::S* __restrict__ partials) {
  using S = typename MON::S;
  using UT = typename UOf<W>::type;
  constexpr int E = 16 / W;
  constexpr int UL = ord_ul(W);
  constexpr bool IS_PROD = std::is_same<MON, ProdMonoid>::value;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per_wave = tile / (AGG_BLOCK / 64);
  const int64_t begin = (int64_t)blockIdx.x * tile + wave * per_wave;
  const int64_t end = begin + per_wave < nvec ? begin + per_wave : nvec;
  const int64_t last_word = HAS_VALID ? ((valid.off + len - 1) >> 6) : 0;
  S wacc = MON::identity();
  for (int64_t g0 = begin; g0 < end; g0 += 64 * UL) {
    const int64_t g = g0 + (int64_t)lane * UL;
    S x = MON::identity();
    if (g < end) {
      const int64_t r0 = g * E - skip;
      const bool interior = g + UL <= end && r0 >= 0 && r0 + UL * E <= len;
      uint4 q[UL];
#pragma unroll
      for (int j = 0; j < UL; ++j) q[j] = (g + j < end) ? base[g + j] : uint4{0, 0, 0, 0};
      uint64_t bits = ~0ull;
      if (interior) {
        if (HAS_VALID) {  // the UL*E <= 64 validity bits of this lane's rows, one funnel shift
          const int64_t pos = valid.off + r0;
          const int64_t wi = pos >> 6;
          const int sh = (int)(pos & 63);
          bits = valid.words[wi] >> sh;
          if (sh) bits |= valid.words[wi < last_word ? wi + 1 : last_word] << (64 - sh);
        }
      } else {
        bits = 0;
#pragma unroll
        for (int i = 0; i < UL * E; ++i) {
          const int64_t r = r0 + i;
          if (g + i / E < end && r >= 0 && r < len && (!HAS_VALID || bv_get(valid, r))) bits |= 1ull << i;
        }
      }
#pragma unroll
      for (int j = 0; j < UL; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const UT raw = vec_elem<W>(q[j], e);
          const bool ok = (bits >> (j * E + e)) & 1ull;
          x = MON::combine(x, MON::lift(widen<W, SIGNED>(ok ? raw : (UT)(IS_PROD ? 1 : 0))));
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) x = MON::combine(x, shfl_down_struct<S>(x, o));
    if (lane == 0) wacc = MON::combine(wacc, x);
  }
  __shared__ S sm[AGG_BLOCK / 64];
  if (lane == 0) sm[wave] = wacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    S r = sm[0];
#pragma unroll
    for (int w = 1; w < AGG_BLOCK / 64; ++w) r = MON::combine(r, sm[w]);
    partials[blockIdx.x] = r;
  }
}

struct CheckedResult {
  int64_t fail_block;  // -1: no overflow
  int64_t pad;
  i128 acc;            // fail_block < 0: the result; else the exact accumulator entering that tile
};

// type range as 128-bit bounds
struct Range {
  i128 lo, hi;
};

__global__ void sum_walk(const SumS* partials, int n, Range rg, CheckedResult* out) {
  i128 base = 0;
  for (int b = 0; b < n; ++b) {
    const SumS p = partials[b];
    if (base + p.mx > rg.hi || base + p.mn < rg.lo) {
      out->fail_block = b;
      out->acc = base;
      return;
    }
    base += p.s;
  }
  out->fail_block = -1;
  out->acc = base;
}

// does multiplying the exact in-range accumulator `acc` through the run summarised by p overflow?
__device__ bool prod_overflows(i128 acc, const ProdS& p, Range rg, bool is_signed) {
  if (acc == 0) return false;
  if (p.flags & F_SAT) return true;
  const u128 a = acc < 0 ? (u128)(-acc) : (u128)acc;
  const u128 m = a * (u128)p.mag;  // < 2^128: a <= 2^64, mag < 2^64
  if (!is_signed) return m > (u128)rg.hi;
  const u128 bound = (u128)rg.hi + 1;  // 2^(w-1)
  if (m < bound) return false;
  if (m > bound) return true;
  return (p.flags & (acc > 0 ? F_MAXPOS : F_MAXNEG)) != 0;  // some prefix lands on +2^(w-1)
}

__global__ void prod_walk(const ProdS* partials, int n, Range rg, int is_signed, CheckedResult* out) {
  i128 acc = 1;
  for (int b = 0; b < n; ++b) {
    const ProdS p = partials[b];
    if (prod_overflows(acc, p, rg, is_signed)) {
      out->fail_block = b;
      out->acc = acc;
      return;
    }
    if (p.flags & F_ZERO) acc = 0;
    else if (acc != 0) acc = (p.flags & F_NEG) ? -(acc * (i128)p.mag) : acc * (i128)p.mag;
  }
  out->fail_block = -1;
  out->acc = acc;
}

// error path only: one thread re-walks the failing tile in row order to find the exact operands
template <int W, bool SIGNED, bool IS_PROD>
__global__ void locate_kernel(const uint4* base, int64_t skip, int64_t len, BitView valid, int64_t v_begin,
                              int64_t v_end, Range rg, i128 acc, i128* out2) {
  constexpr int E = 16 / W;
  const uint8_t* bytes = (const uint8_t*)base;
  for (int64_t r = v_begin * E - skip; r < v_end * E - skip; ++r) {
    if (r < 0 || r >= len || !bv_get(valid, r)) continue;
    typename UOf<W>::type raw = 0;
    memcpy(&raw, bytes + (r + skip) * W, W);
    const i128 x = widen<W, SIGNED>(raw);
    // |acc|, |x| < 2^64, so the magnitude product fits 128 bits: compare through magnitudes
    bool ovf;
    i128 next = 0;
    if (IS_PROD) {
      const u128 a = acc < 0 ? (u128)(-acc) : (u128)acc, b = x < 0 ? (u128)(-x) : (u128)x;
      const u128 hi_lim = (u128)rg.hi, lo_lim = SIGNED ? (u128)rg.hi + 1 : 0;
      const bool neg = (acc < 0) != (x < 0);
      const u128 m = a * b;
      ovf = neg ? m > lo_lim : m > hi_lim;
      if (!ovf) next = neg ? -(i128)m : (i128)m;
    } else {
      next = acc + x;
      ovf = next > rg.hi || next < rg.lo;
    }
    if (ovf) {
      out2[0] = acc;
      out2[1] = x;
      return;
    }
    acc = next;
  }
  out2[0] = acc;  // unreachable when the walk flagged this tile
  out2[1] = 0;
}

void fmt_i128(i128 v, bool is_signed, char* buf, size_t n) {
  if (is_signed) snprintf(buf, n, "%lld", (long long)(int64_t)v);
  else snprintf(buf, n, "%llu", (unsigned long long)(uint64_t)v);
}

template <int W, bool SIGNED, bool IS_PROD>
ah_status run_checked(ah_context* ctx, const ah_array_view* v, bool has_valid, void* scratch, uint64_t* raw_out) {
  using MON = typename std::conditional<IS_PROD, ProdMonoid, SumMonoid>::type;
  using S = typename MON::S;
  Frame f = make_frame(v->values, W, v->length);
  // contiguous tiles: a multiple of 256*UL vectors so every wave gets whole 64*UL-vector iterations
  int64_t tile = (f.nvec + f.grid - 1) / f.grid;
  tile = (tile + AGG_BLOCK * ord_ul(W) - 1) / (AGG_BLOCK * ord_ul(W)) * (AGG_BLOCK * ord_ul(W));
  f.grid = (int)((f.nvec + tile - 1) / tile);
  S* partials = (S*)scratch;
  CheckedResult* res = (CheckedResult*)((char*)scratch + AGG_MAX_GRID * sizeof(SumS));
  i128* operands = (i128*)(res + 1);
  const BitView bv = make_bitview(v->validity, v->validity_bit_offset);
  Range rg;
  if (SIGNED) {
    rg.lo = -((i128)1 << (W * 8 - 1));
    rg.hi = ((i128)1 << (W * 8 - 1)) - 1;
  } else {
    rg.lo = 0;
    rg.hi = ((i128)1 << (W * 8)) - 1;
  }
  {
    ah_prof_scope ps(ctx, "aggregate_checked");
    if (has_valid)
      hipLaunchKernelGGL((ordered_kernel<W, SIGNED, MON, true>), dim3(f.grid), dim3(AGG_BLOCK), 0, ctx->stream, f.base,
                         f.skip, v->length, bv, f.nvec, tile, partials);
    else
      hipLaunchKernelGGL((ordered_kernel<W, SIGNED, MON, false>), dim3(f.grid), dim3(AGG_BLOCK), 0, ctx->stream,
                         f.base, f.skip, v->length, bv, f.nvec, tile, partials);
  }
  if constexpr (IS_PROD)
    hipLaunchKernelGGL(prod_walk, dim3(1), dim3(1), 0, ctx->stream, partials, f.grid, rg, (int)SIGNED, res);
  else
    hipLaunchKernelGGL(sum_walk, dim3(1), dim3(1), 0, ctx->stream, partials, f.grid, rg, res);
  AH_HIP(ctx, hipGetLastError());
  AH_HIP(ctx, ah_d2h_wait(ctx, ctx->pinned, res, sizeof(CheckedResult)));
  CheckedResult r;
  memcpy(&r, ctx->pinned, sizeof r);
  if (r.fail_block < 0) {
    *raw_out = (uint64_t)r.acc;
    return AH_OK;
  }
  const int64_t vb = r.fail_block * tile, ve = std::min<int64_t>(vb + tile, f.nvec);
  hipLaunchKernelGGL((locate_kernel<W, SIGNED, IS_PROD>), dim3(1), dim3(1), 0, ctx->stream, f.base, f.skip, v->length,
                     has_valid ? bv : BitView{nullptr, 0}, vb, ve, rg, r.acc, operands);
  AH_HIP(ctx, hipGetLastError());
  AH_HIP(ctx, ah_d2h_wait(ctx, ctx->pinned, operands, 2 * sizeof(i128)));
  i128 ops[2];
  memcpy(ops, ctx->pinned, sizeof ops);
  char a[48], b[48];
  fmt_i128(ops[0], SIGNED, a, sizeof a);
  fmt_i128(ops[1], SIGNED, b, sizeof b);
  return ah_fail(ctx, AH_ARITHMETIC_OVERFLOW, "Overflow happened on: %s %s %s", a, IS_PROD ? "*" : "+", b);
}

template <int W> ah_status run_checked_w(ah_context* ctx, bool is_signed, bool prod, const ah_array_view* v, bool hv,
                                         void* s, uint64_t* raw) {
  if (is_signed)
    return prod ? run_checked<W, true, true>(ctx, v, hv, s, raw) : run_checked<W, true, false>(ctx, v, hv, s, raw);
  return prod ? run_checked<W, false, true>(ctx, v, hv, s, raw) : run_checked<W, false, false>(ctx, v, hv, s, raw);
}

}  // namespace

extern "C" ah_status ah_aggregate(ah_context* ctx, ah_agg_op op, const ah_array_view* v, ah_scalar* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  memset(out, 0, sizeof *out);
  out->type = v->type;
  hipSetDevice(ctx->device);
  if (op < AH_AGG_SUM || op > AH_AGG_BIT_XOR) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown aggregate %d", (int)op);
  const ah_type t = v->type;
  const int64_t len = v->length;
  int64_t nulls = 0;
  if (v->validity) {
    nulls = v->null_count;
    if (nulls < 0) {
      int64_t set = 0;
      AH_TRY(ah_count_set_bits(ctx, v->validity, v->validity_bit_offset, len, &set));
      nulls = len - set;
    }
  }
  const bool has_valid = nulls > 0;  // `Some(nulls) if null_count > 0` (aggregate.rs:326)

  if (t == AH_BOOL) {  // min_boolean / max_boolean (:372-457)
    if (op != AH_AGG_MIN && op != AH_AGG_MAX)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "only min/max (bool_and/bool_or) aggregate a BooleanArray");
    if (nulls == len) return AH_OK;
    void* tmp = nullptr;
    AH_TRY(ah_pool_alloc(ctx, ah_bitmap_bytes(len), &tmp));
    const BitView vals = make_bitview(v->values, v->values_bit_offset);
    const BitView vld = make_bitview(has_valid ? v->validity : nullptr, v->validity_bit_offset);
    int64_t hits = 0;
    // min: is there a valid false?  valid & ~value.   max: is there a valid true?  value & valid.
    ah_status st = op == AH_AGG_MIN ? ah_bitmap_op(ctx, BM_ANDNOT, vld, vals, BitView{nullptr, 0}, len,
                                                   (unsigned long long*)tmp, &hits)
                                    : ah_bitmap_op(ctx, BM_AND, vals, vld, BitView{nullptr, 0}, len,
                                                   (unsigned long long*)tmp, &hits);
    ah_pool_free(ctx, tmp);
    AH_TRY(st);
    out->is_valid = 1;
    out->bytes[0] = op == AH_AGG_MIN ? (hits == 0) : (hits > 0);
    return AH_OK;
  }

  const bool is_int = ah_type_is_integer(t), is_float = t == AH_FLOAT32 || t == AH_FLOAT64;
  const bool minmax = op == AH_AGG_MIN || op == AH_AGG_MAX;
  const bool bitop = op >= AH_AGG_BIT_AND;
  if (!(is_int || is_float || (t == AH_FLOAT16 && minmax)))
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "aggregate of %s", ah_type_name(t));
  if (bitop && !is_int) return ah_fail(ctx, AH_INVALID_ARGUMENT, "bitwise aggregates need an integer type");
  if (nulls == len) return AH_OK;  // None (:320-323); also the empty array

  const int w = ah_type_width(t);
  void* scratch = nullptr;
  const size_t sbytes = AGG_MAX_GRID * sizeof(SumS) + 256;
  AH_TRY(ah_pool_alloc(ctx, sbytes, &scratch));
  uint64_t raw = 0;
  ah_status st;
  const bool checked = (op == AH_AGG_SUM_CHECKED || op == AH_AGG_PRODUCT_CHECKED) && is_int;
  if (checked) {
    const bool prod = op == AH_AGG_PRODUCT_CHECKED, sg = ah_type_is_signed(t);
    switch (w) {
      case 1: st = run_checked_w<1>(ctx, sg, prod, v, has_valid, scratch, &raw); break;
      case 2: st = run_checked_w<2>(ctx, sg, prod, v, has_valid, scratch, &raw); break;
      case 4: st = run_checked_w<4>(ctx, sg, prod, v, has_valid, scratch, &raw); break;
      default: st = run_checked_w<8>(ctx, sg, prod, v, has_valid, scratch, &raw); break;
    }
  } else {
    int k;
    switch (op) {
      case AH_AGG_SUM: case AH_AGG_SUM_CHECKED: k = K_SUM; break;  // float add_checked never fails
      case AH_AGG_PRODUCT: case AH_AGG_PRODUCT_CHECKED: k = K_PROD; break;
      case AH_AGG_MIN: k = K_MIN; break;
      case AH_AGG_MAX: k = K_MAX; break;
      case AH_AGG_BIT_AND: k = K_AND; break;
      case AH_AGG_BIT_OR: k = K_OR; break;
      default: k = K_XOR; break;
    }
    if (is_float && !minmax) {
      if (t == AH_FLOAT32)
        st = k == K_SUM ? run_agg<4, float, K_SUM, M_FLOAT>(ctx, v, has_valid, scratch, &raw)
                        : run_agg<4, float, K_PROD, M_FLOAT>(ctx, v, has_valid, scratch, &raw);
      else
        st = k == K_SUM ? run_agg<8, double, K_SUM, M_FLOAT>(ctx, v, has_valid, scratch, &raw)
                        : run_agg<8, double, K_PROD, M_FLOAT>(ctx, v, has_valid, scratch, &raw);
    } else {
      const int m = (is_float || t == AH_FLOAT16) ? M_FLOAT : ah_type_is_signed(t) ? M_SIGNED : M_UNSIGNED;
      switch (w) {
        case 1: st = run_int<1>(ctx, k, m, v, has_valid, scratch, &raw); break;
        case 2: st = run_int<2>(ctx, k, m, v, has_valid, scratch, &raw); break;
        case 4: st = run_int<4>(ctx, k, m, v, has_valid, scratch, &raw); break;
        default: st = run_int<8>(ctx, k, m, v, has_valid, scratch, &raw); break;
      }
    }
  }
  ah_pool_free(ctx, scratch);
  AH_TRY(st);
  out->is_valid = 1;
  memcpy(out->bytes, &raw, (size_t)w);  // truncation to the native width == the wrapping result
  return AH_OK;
}
