// filter_expr.hip — the lazily evaluated predicate: compare -> (Kleene) and / or -> filter with the comparison
// evaluated INSIDE the filter's count pass.
//
// Reference call shape (what an engine writes for `WHERE a < 0 AND b >= 0.0`):
//     let m = and_kleene(&lt(&a, &zero)?, &gt_eq(&b, &zero_f)?)?;      arrow-ord/src/cmp.rs:113,164 -> compare_op :220-382
//     filter(&a, &m)                                                    arrow-arith/src/boolean.rs:60-300; filter.rs:201
// i.e. two compare passes that write bitmaps (collect_bool, cmp.rs:580-611), one bitmap pass, and the filter's own
// count pass that reads them back (FilterBuilder::new, filter.rs:256-273: `prep_null_mask_filter` = values AND
// validity, then true_count).  Five launches and ~27.5 GB for 1e9 rows here.
//
// MI355X design: the predicate is handed over as TERMS (op, lhs, rhs | scalar) joined left to right.  ONE kernel
// streams the operand columns with 16-byte loads (two rows per lane), turns every comparison into wave ballots —
// the ballot IS the predicate word (north_star: "__ballot(pred)") — folds the terms' (value, validity) word pairs on
// the scalar unit with exactly the reference's bit formulas (so nulls propagate identically, including the Kleene
// cases), ANDs value and validity (filter.rs:167-171: a null predicate row selects nothing), stores the 1-bit-per-row
// selection words and the per-1024-row-chunk prefix counts.  From there on it is an ordinary ah_filter_predicate: the
// scatter kernels read the selection words like any materialised mask (0.125 B per row).  Per row the predicate
// side costs the operand bytes once and one bit, instead of operand bytes + 2 x (2 bits written + 2 bits read back)
// + 4 bits read + 2 written + 2 read.
//
// Operand types: the integer types and Float32 / Float64 (IEEE totalOrder, equality = bit equality,
// arrow-array/src/arithmetic.rs:400-410).  Ops: eq, neq, lt, lt_eq, gt, gt_eq.  Joins: and, or, and_kleene, or_kleene.
#include "common.hpp"
#include "filter_internal.hpp"

namespace {

constexpr int EXPR_MAX_TERMS = 4;
constexpr int GROUP_CHUNKS_E = 64;  // chunks per workgroup = per count group (group_shift 6)
constexpr int HALF_STEPS = 4;       // wave steps (128 rows each) whose loads are in flight together

struct ExprTerm {
  const void* l;
  const void* r;
  BitView lv, rv;  // validity; words == nullptr: all valid
  int width;       // 1, 2, 4, 8
  int kind;        // 0 signed, 1 unsigned, 2 float
  int op;          // AH_EQ .. AH_GT_EQ
  int l_scalar, r_scalar;
  int l_vec, r_vec;  // operand pointer is aligned for a 2-element vector load
};
struct ExprArgs {
  ExprTerm t[EXPR_MAX_TERMS];
  int nterms;
  int join[EXPR_MAX_TERMS - 1];
  int64_t len;
  unsigned long long* mask_out;
  uint32_t* chunk_prefix;
  uint32_t* group_total;
};

// raw little-endian element -> a signed 64-bit key whose signed order is the type's order (floats: totalOrder)
template <int W> __device__ __forceinline__ int64_t to_key(uint64_t raw, int kind) {
  if constexpr (W == 8) {
    if (kind == 1) return (int64_t)(raw ^ 0x8000000000000000ull);
    const int64_t b = (int64_t)raw;
    return kind == 2 ? (b ^ (int64_t)((uint64_t)(b >> 63) >> 1)) : b;
  } else if constexpr (W == 4) {
    if (kind == 1) return (int64_t)(uint32_t)raw;
    const int32_t b = (int32_t)(uint32_t)raw;
    return kind == 2 ? (int64_t)(b ^ (int32_t)((uint32_t)(b >> 31) >> 1)) : (int64_t)b;
  } else if constexpr (W == 2) {
    return kind == 1 ? (int64_t)(uint16_t)raw : (int64_t)(int16_t)(uint16_t)raw;
  } else {
    return kind == 1 ? (int64_t)(uint8_t)raw : (int64_t)(int8_t)(uint8_t)raw;
  }
}

template <int W> struct RawT;
template <> struct RawT<1> { using type = uint8_t; };
template <> struct RawT<2> { using type = uint16_t; };
template <> struct RawT<4> { using type = uint32_t; };
template <> struct RawT<8> { using type = uint64_t; };
template <int W> struct alignas(2 * W) Pair { typename RawT<W>::type e[2]; };

// the two rows a lane owns (row, row + 1; row even): raw elements.  Rows at or past `len` re-read an in-range
// element (their bits are masked off later): no branch between the loads of a step.
template <int W> __device__ __forceinline__ void load_pair(const void* base, int64_t row, int64_t len, bool scalar, bool vec,
                                                           uint64_t& x0, uint64_t& x1) {
  using R = typename RawT<W>::type;
  const R* p = (const R*)base;
  if (scalar) {
    x0 = x1 = (uint64_t)p[0];
    return;
  }
  if (vec && row + 1 < len) {
    const Pair<W> v = *(const Pair<W>*)(p + row);
    x0 = (uint64_t)v.e[0];
    x1 = (uint64_t)v.e[1];
  } else if (vec) {  // the pair straddles the end: row is the last row of an odd-length column, or past the end
    x0 = x1 = (uint64_t)p[row < len ? row : len - 1];
  } else {
    const int64_t r0 = row < len ? row : len - 1, r1 = row + 1 < len ? row + 1 : len - 1;
    x0 = (uint64_t)p[r0];
    x1 = (uint64_t)p[r1];
  }
}
__device__ __forceinline__ void load_pair_w(int width, const void* base, int64_t row, int64_t len, bool scalar, bool vec,
                                            uint64_t& x0, uint64_t& x1) {
  switch (width) {  // wave-uniform
    case 8: load_pair<8>(base, row, len, scalar, vec, x0, x1); break;
    case 4: load_pair<4>(base, row, len, scalar, vec, x0, x1); break;
    case 2: load_pair<2>(base, row, len, scalar, vec, x0, x1); break;
    default: load_pair<1>(base, row, len, scalar, vec, x0, x1); break;
  }
}
__device__ __forceinline__ int64_t key_w(int width, uint64_t raw, int kind) {
  switch (width) {
    case 8: return to_key<8>(raw, kind);
    case 4: return to_key<4>(raw, kind);
    case 2: return to_key<2>(raw, kind);
    default: return to_key<1>(raw, kind);
  }
}



// The comparison as BALLOTS: `a < b` and `a == b` compile to one v_cmp each whose result IS the 64-lane mask in a
// scalar register pair; the op is three wave-uniform 64-bit masks over (lt, eq, gt) applied on the scalar unit.
// (The first form built 0 / 1 integers per lane with v_cndmask, combined them on the vector unit and compared the
// result with zero to get the ballot: ~16 vector instructions per value pair instead of 4.)
struct OpMask { uint64_t lt, eq, gt; };
__device__ __forceinline__ OpMask op_mask(int op) {
  OpMask m;
  m.lt = (op == AH_LT || op == AH_LT_EQ || op == AH_NEQ) ? ~0ull : 0ull;
  m.eq = (op == AH_EQ || op == AH_LT_EQ || op == AH_GT_EQ) ? ~0ull : 0ull;
  m.gt = (op == AH_GT || op == AH_GT_EQ || op == AH_NEQ) ? ~0ull : 0ull;
  return m;
}
template <typename KT> __device__ __forceinline__ uint64_t cmp_ballot(const OpMask& m, KT a, KT b) {
  const uint64_t lt = __ballot(a < b), eq = __ballot(a == b);
  return (lt & m.lt) | (eq & m.eq) | (~(lt | eq) & m.gt);
}
// branch-free key of the fast path (FW = 8 or 4): the kind as two wave-uniform masks — floats flip the magnitude bits
// of negative values (totalOrder), unsigned types flip the sign bit — so no select and no branch per value
template <int W> struct KeyT { using type = int64_t; };
template <> struct KeyT<4> { using type = int32_t; };
struct KindMask { uint64_t flt, uns; };
template <int W> __device__ __forceinline__ KindMask kind_mask(int kind) {
  KindMask m;
  if constexpr (W == 8) {
    m.flt = kind == 2 ? 0x7FFFFFFFFFFFFFFFull : 0ull;
    m.uns = kind == 1 ? 0x8000000000000000ull : 0ull;
  } else {
    m.flt = kind == 2 ? 0x7FFFFFFFull : 0ull;
    m.uns = kind == 1 ? 0x80000000ull : 0ull;
  }
  return m;
}
template <int W> __device__ __forceinline__ typename KeyT<W>::type fast_key(uint64_t raw, const KindMask& km) {
  if constexpr (W == 8) {
    const int64_t b = (int64_t)raw;
    return b ^ ((b >> 63) & (int64_t)km.flt) ^ (int64_t)km.uns;
  } else {
    const int32_t b = (int32_t)(uint32_t)raw;
    return b ^ ((b >> 31) & (int32_t)(uint32_t)km.flt) ^ (int32_t)(uint32_t)km.uns;
  }
}
// bit p of a 32-bit half word -> bit 2p of a 64-bit word, on the vector unit (lanes 0..15 of the fold do 16 words at once;
// done per wave step on the scalar unit this was 60 scalar instructions per ballot pair: the CU's one scalar unit was
// the kernel's limit)
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}
__device__ __forceinline__ uint64_t spread32(uint32_t x) {
  return (uint64_t)spread16(x & 0xFFFFu) | ((uint64_t)spread16(x >> 16) << 32);
}

// (the first version of this kernel fetched validity words per step with inlined bv_fetch64's: 16 K instructions, past
// the instruction cache, 3.6 TB/s.  Validity is now read once per chunk, 16 words at a time in lanes 0..15.)
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

// The comparison ballots of one chunk (1024 rows at row0) for every term: 8 wave steps of 128 rows; s_raw[k][step] =
// (ballot of rows 2l, ballot of rows 2l + 1), interleaved into value words by the fold.
// FULL: the chunk lies wholly inside the column — unconditional 16-byte loads, HS steps' worth issued back to back.
template <int NT, int FW, bool RS, bool FULL>
__device__ __forceinline__ void expr_value_words(const ExprArgs& a, int64_t row0, int lane, const OpMask (&om)[NT],
                                                 const KindMask (&km)[NT], const int64_t (&rkey)[NT], ulonglong2 (*s_raw)[8]) {
  constexpr int HS = FW ? HALF_STEPS : 1;  // steps whose loads are in flight together
  constexpr int NH = 8 / HS;
  constexpr int W = FW ? FW : 8;
  using R = typename RawT<W>::type;
  using KT = typename KeyT<W>::type;
#pragma unroll 1
  for (int half = 0; half < NH; ++half) {
    uint64_t lx[NT][HS][2], rx[NT][HS][2];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const ExprTerm& tm = a.t[k];
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        const int64_t row = row0 + (half * HS + s) * 128 + 2 * lane;
        if constexpr (FW != 0 && FULL) {
          const Pair<W> lv = *(const Pair<W>*)((const R*)tm.l + row);
          lx[k][s][0] = (uint64_t)lv.e[0], lx[k][s][1] = (uint64_t)lv.e[1];
          if constexpr (!RS) {
            const Pair<W> rv = *(const Pair<W>*)((const R*)tm.r + row);
            rx[k][s][0] = (uint64_t)rv.e[0], rx[k][s][1] = (uint64_t)rv.e[1];
          } else {
            rx[k][s][0] = rx[k][s][1] = 0;
          }
        } else if constexpr (FW != 0) {
          load_pair<W>(tm.l, row, a.len, false, true, lx[k][s][0], lx[k][s][1]);
          if constexpr (!RS) load_pair<W>(tm.r, row, a.len, false, true, rx[k][s][0], rx[k][s][1]);
          else rx[k][s][0] = rx[k][s][1] = 0;
        } else {
          load_pair_w(tm.width, tm.l, row, a.len, tm.l_scalar != 0, tm.l_vec != 0, lx[k][s][0], lx[k][s][1]);
          load_pair_w(tm.width, tm.r, row, a.len, tm.r_scalar != 0, tm.r_vec != 0, rx[k][s][0], rx[k][s][1]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < HS; ++s) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const ExprTerm& tm = a.t[k];
        uint64_t b0, b1;  // bit l = row 2l (b0) / 2l + 1 (b1)
        if constexpr (FW != 0) {
          const KT r0 = RS ? (KT)rkey[k] : fast_key<W>(rx[k][s][0], km[k]);
          const KT r1 = RS ? (KT)rkey[k] : fast_key<W>(rx[k][s][1], km[k]);
          b0 = cmp_ballot<KT>(om[k], fast_key<W>(lx[k][s][0], km[k]), r0);
          b1 = cmp_ballot<KT>(om[k], fast_key<W>(lx[k][s][1], km[k]), r1);
        } else {
          b0 = cmp_ballot<int64_t>(om[k], key_w(tm.width, lx[k][s][0], tm.kind), key_w(tm.width, rx[k][s][0], tm.kind));
          b1 = cmp_ballot<int64_t>(om[k], key_w(tm.width, lx[k][s][1], tm.kind), key_w(tm.width, rx[k][s][1], tm.kind));
        }
        if (lane == 0) s_raw[k][half * HS + s] = make_ulonglong2(b0, b1);
      }
    }
  }
}

// One workgroup = 64 chunks (65 536 rows); in iteration `it` wave w owns chunk it * 4 + w: 8 wave steps of 128 rows,
// lane l of a step owning rows 2l and 2l + 1 (one 16-byte load per 8-byte operand).  Per step and term: 2 ballots ->
// 2 value words (bit-interleaved on the scalar unit); the terms' validity words are folded once per chunk, 16 words at
// a time in lanes 0..15.
// FW != 0: the fast instantiation — every term compares FW-byte operands, every array operand pointer is aligned for a
// two-element vector load, and (RS) every right-hand side is a scalar: the loads of four steps are unconditional vector
// loads issued back to back and the comparison is branch-free.  FW == 0: any mix of widths / alignments / scalar sides,
// dispatched per term at run time (correct, not tuned).
#ifdef AH_EXPR_WAVES  // ablation builds: force the occupancy of the count pass (profiles/r05_expr_occupancy.md)
#define AH_EXPR_ATTR __attribute__((amdgpu_waves_per_eu(AH_EXPR_WAVES, 8)))
#else
#define AH_EXPR_ATTR
#endif
template <int NT, int FW, bool RS>
__global__ void __launch_bounds__(256) AH_EXPR_ATTR filter_expr_count_kernel(ExprArgs a) {
  __shared__ uint32_t s_cnt[GROUP_CHUNKS_E];
  __shared__ ulonglong2 s_raw[4][NT][8];  // per wave: the comparison ballots of the chunk's terms, per wave step
  const int t = threadIdx.x, lane = t & 63, wave = ah_uniform(t >> 6);
  const int64_t chunk_base = (int64_t)blockIdx.x * GROUP_CHUNKS_E;
  const int64_t nchunks = (a.len + AH_FILTER_CHUNK_ROWS - 1) / AH_FILTER_CHUNK_ROWS;
  OpMask om[NT];
  KindMask km[NT];
  int64_t rkey[NT];  // RS: the scalar right-hand sides, once
  uint64_t lsw[NT], rsw[NT];  // validity of scalar operands as a word, once (a load here waits for nothing else)
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    om[k] = op_mask(a.t[k].op);
    km[k] = kind_mask<(FW ? FW : 8)>(a.t[k].kind);
    rkey[k] = 0;
    lsw[k] = (a.t[k].l_scalar && a.t[k].lv.words && !bv_get(a.t[k].lv, 0)) ? 0ull : ~0ull;
    rsw[k] = (a.t[k].r_scalar && a.t[k].rv.words && !bv_get(a.t[k].rv, 0)) ? 0ull : ~0ull;
    lsw[k] = (uint64_t)ah_uniform64((int64_t)lsw[k]);
    rsw[k] = (uint64_t)ah_uniform64((int64_t)rsw[k]);
    if constexpr (FW != 0 && RS) {
      uint64_t x0, x1;
      load_pair<(FW ? FW : 8)>(a.t[k].r, 0, 1, true, false, x0, x1);
      rkey[k] = (int64_t)fast_key<(FW ? FW : 8)>(x0, km[k]);
    }
  }
  for (int it = 0; it < GROUP_CHUNKS_E / 4; ++it) {
    const int64_t chunk = chunk_base + it * 4 + wave;
    if (chunk >= nchunks) {  // wave-uniform
      if (lane == 0) s_cnt[it * 4 + wave] = 0;
      continue;
    }
    const int64_t row0 = ah_uniform64(chunk * AH_FILTER_CHUNK_ROWS);
    // 1. the chunk's validity words, term by term: lane j < 16 holds word j.  Only ISSUED here (no use of the data): the
    //    value loads below go out behind them without a wait in between.
    const int64_t vs = row0 + ((int64_t)lane << 6), vsc = vs < a.len ? vs : 0;
    BvRaw rl[NT] = {}, rr[NT] = {};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (a.t[k].lv.words && !a.t[k].l_scalar && lane < 16) rl[k] = bv_issue(a.t[k].lv, vsc, a.len);
      if (a.t[k].rv.words && !a.t[k].r_scalar && lane < 16) rr[k] = bv_issue(a.t[k].rv, vsc, a.len);
    }
    // 2. the value words: ballots of the comparisons, four steps' loads in flight at a time.  A chunk that lies wholly
    //    inside the column (all but the last) loads without a bounds test: the tested form compiles to a divergent
    //    branch per load with an s_waitcnt vmcnt(0) behind it — ONE load in flight per wave (5.3 TB/s for 1e9 rows).
    if (FW != 0 && row0 + AH_FILTER_CHUNK_ROWS <= a.len)  // wave-uniform
      expr_value_words<NT, FW, RS, true>(a, row0, lane, om, km, rkey, s_raw[wave]);
    else
      expr_value_words<NT, FW, RS, false>(a, row0, lane, om, km, rkey, s_raw[wave]);
    // 3. fold the terms, 16 words at a time (lane j < 16 = word j).  Bit formulas of arrow-arith/src/boolean.rs.
    //    (same-wave LDS traffic: the writes above are ordered before these reads by the wave's own program order)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    uint64_t v = 0, nn = 0;
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const uint64_t lw = a.t[k].l_scalar ? lsw[k] : (a.t[k].lv.words ? bv_finish(rl[k], vsc, a.len) : ~0ull);
        const uint64_t rw = a.t[k].r_scalar ? rsw[k] : (a.t[k].rv.words ? bv_finish(rr[k], vsc, a.len) : ~0ull);
        // word j of the chunk = rows 64j .. 64j + 63 = half (j & 1) of step j >> 1: its two ballot halves, interleaved
        const ulonglong2 bb = s_raw[wave][k][lane >> 1];
        const int sh = (lane & 1) * 32;
        const uint64_t tv = spread32((uint32_t)(bb.x >> sh)) | (spread32((uint32_t)(bb.y >> sh)) << 1), tk = lw & rw;
        if (k == 0) {
          v = tv, nn = tk;
        } else {
          const int j = a.join[k - 1];  // uniform
          if (j == AH_BOOL_AND) {  // :279 binary_boolean_kernel: values a & b, nulls = union of the null sets
            v &= tv, nn &= tk;
          } else if (j == AH_BOOL_OR) {  // :300
            v |= tv, nn &= tk;
          } else if (j == AH_BOOL_AND_KLEENE) {  // :60-150: valid where both are, or where either side is a valid false
            const uint64_t m = (nn & tk) | (nn & ~v) | (tk & ~tv);
            v &= tv, nn = m;
          } else {  // AH_BOOL_OR_KLEENE :152-241: valid where both are, or where either side is a valid true
            const uint64_t m = (nn & tk) | (nn & v) | (tk & tv);
            v |= tv, nn = m;
          }
        }
      }
    }
    // prep_null_mask_filter (filter.rs:167-171): selected = value AND valid; rows past len are not rows
    uint64_t sel = v & nn;
    const int64_t rem = a.len - (row0 + ((int64_t)lane << 6));
    if (rem < 64) sel = rem <= 0 ? 0 : (sel & ((1ull << rem) - 1));
    if (lane >= 16) sel = 0;
    if (lane < 16) a.mask_out[chunk * 16 + lane] = sel;  // one 128-byte store per chunk
    int c = __popcll(sel);
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 8, 64);
    if (lane == 0) s_cnt[it * 4 + wave] = (uint32_t)c;
  }
  __syncthreads();
  if (wave == 0) {
    const int v = (int)s_cnt[lane];
    const int incl = wave_scan_incl(v);
    if (chunk_base + lane < nchunks) a.chunk_prefix[chunk_base + lane] = (uint32_t)(incl - v);
    if (lane == 63) a.group_total[blockIdx.x] = (uint32_t)incl;
  }
}

template <int NT>
void launch_expr_count(ah_context* ctx, const ExprArgs& a, int64_t ngroups, int fw, bool rs) {
  const dim3 g((unsigned)ngroups), b(256);
  if (fw == 8 && rs) filter_expr_count_kernel<NT, 8, true><<<g, b, 0, ctx->stream>>>(a);
  else if (fw == 8) filter_expr_count_kernel<NT, 8, false><<<g, b, 0, ctx->stream>>>(a);
  else if (fw == 4 && rs) filter_expr_count_kernel<NT, 4, true><<<g, b, 0, ctx->stream>>>(a);
  else if (fw == 4) filter_expr_count_kernel<NT, 4, false><<<g, b, 0, ctx->stream>>>(a);
  else filter_expr_count_kernel<NT, 0, false><<<g, b, 0, ctx->stream>>>(a);
}

const char* cmp_sym_e(int op) {
  switch (op) {
    case AH_EQ: return "==";
    case AH_NEQ: return "!=";
    case AH_LT: return "<";
    case AH_LT_EQ: return "<=";
    case AH_GT: return ">";
    default: return ">=";
  }
}

}  // namespace

// the terms of a filter expression, validated with compare_op's / binary_boolean_kernel's texts
static ah_status parse_terms(ah_context* ctx, int32_t n_terms, const ah_filter_term* terms, const ah_boolean_op* joins, ExprArgs* pa,
                             int64_t* plen) {
  if (!terms || (n_terms > 1 && !joins)) return AH_INVALID_ARGUMENT;
  if (n_terms < 1 || n_terms > EXPR_MAX_TERMS)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "a filter expression takes 1..%d comparison terms, got %d", EXPR_MAX_TERMS, n_terms);
  ExprArgs& a = *pa;
  a = ExprArgs{};
  a.nterms = n_terms;
  int64_t len = -1;
  for (int k = 0; k < n_terms; ++k) {
    const ah_filter_term& tm = terms[k];
    if (!tm.lhs || !tm.rhs) return AH_INVALID_ARGUMENT;
    const bool ls = tm.lhs_is_scalar != 0, rs = tm.rhs_is_scalar != 0;
    if (tm.op < AH_EQ || tm.op > AH_GT_EQ)
      return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "filter expression: comparison op %d (eq, neq, lt, lt_eq, gt, gt_eq only)", tm.op);
    // compare_op (cmp.rs:228-264)
    if (tm.lhs->length != tm.rhs->length && !ls && !rs)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "Cannot compare arrays of different lengths, got %lld vs %lld",
                     (long long)tm.lhs->length, (long long)tm.rhs->length);
    if (tm.lhs->type != tm.rhs->type)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid comparison operation: %s %s %s", ah_type_name(tm.lhs->type),
                     cmp_sym_e(tm.op), ah_type_name(tm.rhs->type));
    const ah_type ty = tm.lhs->type;
    if (!(ah_type_is_integer(ty) || ah_type_is_float(ty)))
      return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "filter expression: comparison of %s operands (integers and Float32 / Float64 only)",
                     ah_type_name(ty));
    if ((ls && tm.lhs->length < 1) || (rs && tm.rhs->length < 1))
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar datum must have length 1");
    if (ls && rs) return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "filter expression: a term needs at least one array operand");
    const int64_t tl = ls ? tm.rhs->length : tm.lhs->length;
    if (len < 0) len = tl;
    // binary_boolean_kernel (boolean.rs:262-266)
    if (tl != len) return ah_fail(ctx, AH_COMPUTE_ERROR, "Cannot perform bitwise operation on arrays of different length");
    if (k > 0) {
      const int j = joins[k - 1];
      if (j != AH_BOOL_AND && j != AH_BOOL_OR && j != AH_BOOL_AND_KLEENE && j != AH_BOOL_OR_KLEENE)
        return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "filter expression: join op %d (and, or, and_kleene, or_kleene only)", j);
      a.join[k - 1] = j;
    }
    ExprTerm& e = a.t[k];
    e.l = tm.lhs->values, e.r = tm.rhs->values;
    e.width = ah_type_width(ty);
    e.kind = ah_type_is_float(ty) ? 2 : (ah_type_is_signed(ty) ? 0 : 1);
    e.op = tm.op;
    e.l_scalar = ls, e.r_scalar = rs;
    auto valid_of = [](const ah_array_view* v) {
      return (v->validity && v->null_count != 0) ? make_bitview(v->validity, v->validity_bit_offset) : BitView{nullptr, 0};
    };
    e.lv = valid_of(tm.lhs), e.rv = valid_of(tm.rhs);
    e.l_vec = !ls && tl >= 2 && ((uintptr_t)e.l % (2 * e.width)) == 0;
    e.r_vec = !rs && tl >= 2 && ((uintptr_t)e.r % (2 * e.width)) == 0;
  }
  *plen = len;
  return AH_OK;
}

extern "C" ah_status ah_filter_predicate_build_expr(ah_context* ctx, int32_t n_terms, const ah_filter_term* terms,
                                                    const ah_boolean_op* joins, ah_filter_predicate** out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out) return AH_INVALID_ARGUMENT;
  *out = nullptr;
  hipSetDevice(ctx->device);
  ExprArgs a{};
  int64_t len = -1;
  AH_TRY(parse_terms(ctx, n_terms, terms, joins, &a, &len));
  auto* p = new ah_filter_predicate();
  p->len = len;
  p->mask_valid = BitView{nullptr, 0};
  p->group_shift = 6;
  if (len <= 0) {
    p->len = 0;
    *out = p;
    return AH_OK;
  }
  const int64_t nchunks = ah_ceil_div(len, AH_FILTER_CHUNK_ROWS), ngroups = ah_ceil_div(nchunks, GROUP_CHUNKS_E);
  const size_t b_mask = (size_t)nchunks * 16 * 8;
  const size_t b_chunk = ((size_t)nchunks * 4 + 255) & ~(size_t)255;
  const size_t b_gt = ((size_t)ngroups * 4 + 255) & ~(size_t)255;
  const size_t b_gp = ((size_t)ngroups * 8 + 255) & ~(size_t)255;
  ah_status st = ah_pool_alloc(ctx, b_mask + b_chunk + b_gt + b_gp + 256, &p->block);
  if (st != AH_OK) {
    delete p;
    return st;
  }
  char* base = (char*)p->block;
  a.mask_out = (unsigned long long*)base;
  p->mask = BitView{(const uint64_t*)base, 0};
  p->chunk_prefix = (uint32_t*)(base + b_mask);
  uint32_t* group_total = (uint32_t*)(base + b_mask + b_chunk);
  p->group_prefix = (unsigned long long*)(base + b_mask + b_chunk + b_gt);
  p->total_dev = (unsigned long long*)(base + b_mask + b_chunk + b_gt + b_gp);
  a.len = len;
  a.chunk_prefix = p->chunk_prefix;
  a.group_total = group_total;
  const uint64_t seq = ah_mail_next(ctx);
  {
    ah_prof_scope ps(ctx, "filter_expr_count");
    // the fast instantiations: one operand width (8 or 4), every array operand vector-aligned, left sides arrays, right
    // sides all scalars (RS) or all arrays
    int fw = a.t[0].width;
    bool all_rs = true, all_ra = true;
    for (int k = 0; k < n_terms; ++k) {
      const ExprTerm& e = a.t[k];
      if (e.width != fw || e.l_scalar || !e.l_vec || (!e.r_scalar && !e.r_vec)) fw = 0;
      all_rs = all_rs && e.r_scalar;
      all_ra = all_ra && !e.r_scalar;
    }
    if ((fw != 8 && fw != 4) || !(all_rs || all_ra)) fw = 0;
    static const char* force = getenv("AH_FILTER_EXPR_GENERIC");  // tests: run the generic instantiation on any input
    if (force && force[0] == '1') fw = 0;
    switch (n_terms) {
      case 1: launch_expr_count<1>(ctx, a, ngroups, fw, all_rs); break;
      case 2: launch_expr_count<2>(ctx, a, ngroups, fw, all_rs); break;
      case 3: launch_expr_count<3>(ctx, a, ngroups, fw, all_rs); break;
      default: launch_expr_count<4>(ctx, a, ngroups, fw, all_rs); break;
    }
  }
  ah_filter_launch_group_scan(ctx, group_total, ngroups, p->group_prefix, p->total_dev, 0, seq);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  if (e != hipSuccess) {
    ah_pool_free(ctx, p->block);
    delete p;
    return ah_fail(ctx, AH_HIP_ERROR, "filter expression count failed: %s", hipGetErrorString(e));
  }
  p->count = (int64_t)ctx->pinned[0];
  *out = p;
  return AH_OK;
}

// `filter(values, <expression>)` in one call: the lazy predicate, then the ordinary scatter.
// (A SINGLE-PASS form — evaluate, decoupled look-back over ticket-ordered tiles with 8-byte state granules, compact, every
// operand read once, 18.2 GB instead of 26.7 GB per 1e9 rows — was built, was bit-exact at 1e9 rows, and lost: 6.7 ms
// against 4.6 ms for these two passes, 5.6 ms even with the look-back ablated.  Holding two tiles' values in registers
// across the ticket, the aggregates and the look-back leaves 3 workgroups per CU working in phases that each expose a full
// memory latency, where the two streaming kernels run at 5.8-5.9 TB/s.  It also needed the output allocated for the worst
// case (8 GB here).  A stash form — the count pass also leaves each chunk's selected values compacted in a slot, the
// second pass copies slots instead of re-reading the column: 21.8 GB — was built too and lost as well (5.2 ms against 5.0 on
// the same box: ranking + compacting inside the count pass costs more than the re-read saves).  Numbers for both:
// profiles/r03_single_pass_lookback.md; nothing of either is in the tree.)
extern "C" ah_status ah_filter_expr(ah_context* ctx, int32_t n_terms, const ah_filter_term* terms, const ah_boolean_op* joins,
                                    const ah_array_view* values, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  ah_filter_predicate* p = nullptr;
  AH_TRY(ah_filter_predicate_build_expr(ctx, n_terms, terms, joins, &p));
  const ah_status st = ah_filter_predicate_apply(ctx, p, values, out);
  ah_filter_predicate_free(ctx, p);
  return st;
}

// Buffer::shrink_to_fit for a result whose buffers were allocated for the worst case (the one-launch filter of small
// batches, filter_small.hip): values / validity are copied into exact-size allocations.  Fixed-width
// results only; a result that already fits, or borrows its buffers, is left as it is.
extern "C" ah_status ah_array_shrink_to_fit(ah_context* ctx, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out) return AH_INVALID_ARGUMENT;
  const int w = ah_type_width(out->type);
  if (w <= 0 || out->type == AH_BOOL || (out->flags & (AH_OUT_BORROWED | AH_OUT_BORROWED_VALUES)) || !out->values) return AH_OK;
  hipSetDevice(ctx->device);
  const size_t need = (size_t)out->length * w, bneed = ah_bitmap_bytes(out->length);
  void *nv = nullptr, *nb = nullptr;
  const bool sv = (size_t)out->values_bytes > need + (1u << 20);
  const bool sb = out->validity && (size_t)out->validity_bytes > bneed + (1u << 20);
  if (!sv && !sb) return AH_OK;
  ah_status st = AH_OK;
  if (sv) st = ah_out_alloc(ctx, need, &nv);
  if (st == AH_OK && sb) st = ah_out_alloc(ctx, bneed, &nb);
  hipError_t e = hipSuccess;
  if (st == AH_OK && sv) e = hipMemcpyAsync(nv, out->values, need, hipMemcpyDeviceToDevice, ctx->stream);
  if (st == AH_OK && e == hipSuccess && sb) e = hipMemcpyAsync(nb, out->validity, bneed, hipMemcpyDeviceToDevice, ctx->stream);
  if (st == AH_OK && e == hipSuccess) e = ah_stream_wait(ctx);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, nv, need);
    ah_out_free(ctx, nb, bneed);
    return st != AH_OK ? st : ah_fail(ctx, AH_HIP_ERROR, "shrink_to_fit failed: %s", hipGetErrorString(e));
  }
  if (sv) {
    ah_out_free(ctx, out->values, (size_t)out->values_bytes);
    out->values = nv;
    out->values_bytes = (int64_t)need;
  }
  if (sb) {
    ah_out_free(ctx, out->validity, (size_t)out->validity_bytes);
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bneed;
  }
  return AH_OK;
}
