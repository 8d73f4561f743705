// arith.hip — arrow_arith::numeric on MI355X.
//
// Reference path: add/add_wrapping/... (arrow-arith/src/numeric.rs:36-81) ->
// arithmetic_op :225 -> integer_op :328 / float_op :357 -> op!/try_op! :278-317
// -> arity::binary (arity.rs:104-135: every slot evaluated, nulls =
// NullBuffer::union) or arity::try_binary (arity.rs:254-299: valid slots only,
// zero elsewhere, first error aborts) / PrimitiveArray::{unary,try_unary}.
//
// MI355X design: the validity union is a separate word-parallel bitmap kernel
// (1.5% of the bytes) that also yields null_count; the value kernel is a pure
// 16-byte-per-lane stream (HBM roofline: 3 x width bytes per row).  Checked ops
// read the freshly produced union words (offset 0) to skip null slots and report
// the first failing row through atomicMin so the host can reproduce the
// reference's error text ("Overflow happened on: l op r", "Divide by zero error").
#include "common.hpp"

#include <limits>
#include <type_traits>

namespace {

enum { OP_ADD = 0, OP_ADD_W, OP_SUB, OP_SUB_W, OP_MUL, OP_MUL_W, OP_DIV, OP_REM, OP_NEG, OP_NEG_W,
       // arrow_arith::bitwise (arrow-arith/src/bitwise.rs:42-135), integers only
       OP_BAND, OP_BOR, OP_BXOR, OP_SHL, OP_SHR, OP_BANDNOT, OP_BNOT };
constexpr bool op_is_unary(int op) { return op == OP_NEG || op == OP_NEG_W || op == OP_BNOT; }

const char* op_sym(int op) {  // Display for Op (numeric.rs:203-213)
  switch (op) {
    case OP_ADD: case OP_ADD_W: return "+";
    case OP_SUB: case OP_SUB_W: return "-";
    case OP_MUL: case OP_MUL_W: return "*";
    case OP_DIV: return "/";
    case OP_REM: return "%";
    case OP_BAND: return "&";
    case OP_BOR: return "|";
    case OP_BXOR: return "^";
    case OP_SHL: return "<<";
    case OP_SHR: return ">>";
    case OP_BANDNOT: return "&!";
    default: return "%";
  }
}

// NaN bit patterns as the reference's host produces them.  The reference runs on x86-64 (SSE2), and its
// float kernels compare and sort NaNs by their BITS (total_cmp / to_bits equality,
// arrow-array/src/arithmetic.rs:400-410), so the sign and payload of a NaN result are observable:
//   * an operation that GENERATES a NaN (inf - inf, 0 * inf, 0 / 0, inf / inf, x % 0, inf % y) yields the x86
//     "QNaN floating-point indefinite" 0xFFF8000000000000 / 0xFFC00000 — CDNA4 alone would give 0x7FF8... /
//     0x7FC0..., and `lt(add(inf, -inf), 0.0)` would flip;
//   * a NaN operand is PROPAGATED: the result is that operand with the quiet bit set, the left one first
//     (SSE rule: "first source operand if it is a NaN, else the second").
template <typename T> __device__ __forceinline__ T x86_default_nan();
template <> __device__ __forceinline__ double x86_default_nan<double>() { return __longlong_as_double((long long)0xFFF8000000000000ull); }
template <> __device__ __forceinline__ float x86_default_nan<float>() { return __uint_as_float(0xFFC00000u); }
__device__ __forceinline__ double quieted(double x) { return __longlong_as_double(__double_as_longlong(x) | 0x0008000000000000ll); }
__device__ __forceinline__ float quieted(float x) { return __uint_as_float(__float_as_uint(x) | 0x00400000u); }

// element semantics: ArrowNativeTypeOp (arrow-array/src/arithmetic.rs:147-430)
template <typename T, int OP>
__device__ __forceinline__ bool apply(T l, T r, T* out) {  // returns false on error
  if constexpr (std::is_same<T, ah_f16>::value) {
    // half::f16 (numeric.rs:240 float_op::<Float16Type>): to f32, ONE f32 operation with the same NaN rules, one rounding back;
    // Neg flips bit 15 of the pattern, NaNs included
    if constexpr (op_is_unary(OP)) {
      out->bits = (uint16_t)(l.bits ^ 0x8000u);
    } else {
      float o;
      apply<float, OP>(ah_f16_to_f32(l), ah_f16_to_f32(r), &o);
      *out = ah_f32_to_f16(o);
    }
    return true;
  } else if constexpr (std::is_floating_point<T>::value) {
    T o;
    if constexpr (OP == OP_ADD || OP == OP_ADD_W) o = l + r;
    else if constexpr (OP == OP_SUB || OP == OP_SUB_W) o = l - r;
    else if constexpr (OP == OP_MUL || OP == OP_MUL_W) o = l * r;
    else if constexpr (OP == OP_DIV) o = l / r;
    else if constexpr (OP == OP_REM) o = fmod(l, r);
    else o = -l;  // neg: a sign flip, also of a NaN (xorps on the reference's host)
    if constexpr (!op_is_unary(OP)) {
      if (l != l) o = quieted(l);
      else if (r != r) o = quieted(r);
      else if (o != o) o = x86_default_nan<T>();
    }
    *out = o;
    return true;
  } else {
    using U = typename std::make_unsigned<T>::type;
    if constexpr (OP == OP_ADD) return !__builtin_add_overflow(l, r, out);
    else if constexpr (OP == OP_SUB) return !__builtin_sub_overflow(l, r, out);
    else if constexpr (OP == OP_MUL) return !__builtin_mul_overflow(l, r, out);
    else if constexpr (OP == OP_ADD_W) { *out = (T)((U)l + (U)r); return true; }
    else if constexpr (OP == OP_SUB_W) { *out = (T)((U)l - (U)r); return true; }
    else if constexpr (OP == OP_MUL_W) { *out = (T)((U)l * (U)r); return true; }
    else if constexpr (OP == OP_DIV) {
      if (r == 0) return false;
      if (std::is_signed<T>::value && l == std::numeric_limits<T>::min() && r == (T)-1) return false;
      *out = (T)(l / r);
      return true;
    } else if constexpr (OP == OP_REM) {  // numeric.rs:345-351
      if (r == 0) return false;
      if (std::is_signed<T>::value && r == (T)-1) *out = 0;
      else *out = (T)(l % r);
      return true;
    } else if constexpr (OP == OP_NEG) {
      if (std::is_signed<T>::value && l == std::numeric_limits<T>::min()) return false;
      *out = (T)(0 - (U)l);
      return true;
    } else if constexpr (OP == OP_BAND) { *out = (T)(l & r); return true; }
    else if constexpr (OP == OP_BOR) { *out = (T)(l | r); return true; }
    else if constexpr (OP == OP_BXOR) { *out = (T)(l ^ r); return true; }
    else if constexpr (OP == OP_BANDNOT) { *out = (T)(l & ~r); return true; }
    else if constexpr (OP == OP_BNOT) { *out = (T)~l; return true; }
    else if constexpr (OP == OP_SHL) {  // wrapping_shl(b as usize as u32): the count is taken modulo the bit width
      *out = (T)((U)l << ((unsigned)(U)r & (sizeof(T) * 8 - 1)));
      return true;
    } else if constexpr (OP == OP_SHR) {  // wrapping_shr: arithmetic for signed types
      *out = (T)(l >> ((unsigned)(U)r & (sizeof(T) * 8 - 1)));
      return true;
    } else {  // OP_NEG_W
      *out = (T)(0 - (U)l);
      return true;
    }
  }
}

template <typename T, int V> struct alignas(sizeof(T) * V) VecT { T e[V]; };

struct ArithArgs {
  const void* l;
  const void* r;
  void* out;
  int64_t len;
  int l_scalar, r_scalar;
  const unsigned long long* valid;  // union words (offset 0) for checked ops, else nullptr
  unsigned long long* first_err;
  // Small inputs (grid <= 2048 blocks), unchecked ops: the kernel also writes the result's validity words (va & vb) and
  // accumulates their popcount — the call is this launch + the read-back kernel (it was the value kernel, a bitmap
  // kernel, a sum kernel and a copy kernel: 20 us at 10^4 rows).
  int post;
  BitView va, vb;             // post: the operands' validity (words == nullptr: all valid)
  unsigned long long* vout;   // post: result validity words
  unsigned long long* total;  // post: scratch word the popcount accumulates in
};

// CHECKED: evaluate valid slots only, zero elsewhere (try_binary / try_unary)
template <typename T, int OP, int V, bool CHECKED>
__global__ void __launch_bounds__(256) arith_kernel(ArithArgs a) {
  using VT = VecT<T, V>;
  const T* lp = (const T*)a.l;
  const T* rp = (const T*)a.r;
  T* op = (T*)a.out;
  T ls = a.l_scalar ? lp[0] : T{};
  T rs = a.r_scalar ? rp[0] : T{};
  unsigned long long err = ~0ull;
  const int64_t nvec = (a.len + V - 1) / V;
  constexpr int U = 4;
  for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 1; base < nvec;
       base += (int64_t)gridDim.x * 256 * U) {
    VT lv[U], rv[U];
    // issue all loads of the unrolled group first
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t vi = base + (int64_t)u * gridDim.x * 256;
      if (vi < nvec) {
        int64_t i = vi * V;
        if (i + V <= a.len) {
          if (!a.l_scalar) lv[u] = ah_ld_stream<ah_nt_l(false)>((const VT*)(lp + i));
          if (!op_is_unary(OP) && !a.r_scalar) rv[u] = ah_ld_stream<ah_nt_l(false)>((const VT*)(rp + i));
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            lv[u].e[e] = (!a.l_scalar && i + e < a.len) ? lp[i + e] : T{};
            rv[u].e[e] = (!op_is_unary(OP) && !a.r_scalar && i + e < a.len) ? rp[i + e] : T{};
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t vi = base + (int64_t)u * gridDim.x * 256;
      if (vi >= nvec) continue;
      int64_t i = vi * V;
      VT ov;
      uint32_t vbits = 0xFFFFFFFFu;
      if constexpr (CHECKED) {
        if (a.valid) vbits = (uint32_t)(a.valid[i >> 6] >> (i & 63));
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        T x = a.l_scalar ? ls : lv[u].e[e];
        T y = a.r_scalar ? rs : rv[u].e[e];
        T o = T{};
        if (!CHECKED || ((vbits >> e) & 1u)) {
          bool ok = apply<T, OP>(x, y, &o);
          if (CHECKED && !ok && i + e < a.len) {
            unsigned long long pos = (unsigned long long)(i + e);
            err = pos < err ? pos : err;
            o = T{};
          }
        }
        ov.e[e] = o;
      }
      if (i + V <= a.len) ah_st_stream<ah_nt_s(false)>((VT*)(op + i), ov);
      else
        for (int e = 0; e < V; ++e) if (i + e < a.len) op[i + e] = ov.e[e];
    }
  }
  if constexpr (CHECKED) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long other = __shfl_xor(err, o, 64);
      err = other < err ? other : err;
    }
    if ((threadIdx.x & 63) == 0 && err != ~0ull) atomicMin(a.first_err, err);
  }
  if (a.post) {  // uniform
    unsigned long long acc = 0;
    {
      const int64_t nwords = (a.len + 63) >> 6;
      for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
        uint64_t r = bv_fetch64(a.va, w << 6, a.len);
        if (a.vb.words) r &= bv_fetch64(a.vb, w << 6, a.len);
        a.vout[w] = r;
        acc += __popcll(r);
      }
    }
    acc = wave_reduce_add64(acc);
    __shared__ unsigned long long sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ah_count_add(a.total, sm[0] + sm[1] + sm[2] + sm[3]);
  }
}

template <typename T, int OP, bool CHECKED>
void launch_arith_op(ah_context* ctx, const ArithArgs& a, bool aligned) {
  constexpr int V = 16 / sizeof(T);
  int64_t nvec = ah_ceil_div(a.len, aligned ? V : 1);
  // one unrolled pass per workgroup: on MI355X a 2-read/1-write stream peaks with >= 64K
  // short-lived workgroups (5.6 TB/s) rather than a persistent grid-stride grid (5.1 TB/s)
  int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(nvec, 256 * 4), (int64_t)1 << 30));
  if (aligned) arith_kernel<T, OP, V, CHECKED><<<grid, 256, 0, ctx->stream>>>(a);
  else arith_kernel<T, OP, 1, CHECKED><<<grid, 256, 0, ctx->stream>>>(a);
}

template <typename T>
void launch_arith(ah_context* ctx, int op, const ArithArgs& a, bool aligned) {
  constexpr bool F = ah_is_fp<T>::value;
  switch (op) {
    case OP_ADD: launch_arith_op<T, OP_ADD, !F>(ctx, a, aligned); break;
    case OP_ADD_W: launch_arith_op<T, OP_ADD_W, false>(ctx, a, aligned); break;
    case OP_SUB: launch_arith_op<T, OP_SUB, !F>(ctx, a, aligned); break;
    case OP_SUB_W: launch_arith_op<T, OP_SUB_W, false>(ctx, a, aligned); break;
    case OP_MUL: launch_arith_op<T, OP_MUL, !F>(ctx, a, aligned); break;
    case OP_MUL_W: launch_arith_op<T, OP_MUL_W, false>(ctx, a, aligned); break;
    case OP_DIV: launch_arith_op<T, OP_DIV, !F>(ctx, a, aligned); break;
    case OP_REM: launch_arith_op<T, OP_REM, !F>(ctx, a, aligned); break;
    case OP_NEG: launch_arith_op<T, OP_NEG, !F>(ctx, a, aligned); break;
    case OP_NEG_W: launch_arith_op<T, OP_NEG_W, false>(ctx, a, aligned); break;
    default:
      if constexpr (!F) {
        switch (op) {
          case OP_BAND: launch_arith_op<T, OP_BAND, false>(ctx, a, aligned); break;
          case OP_BOR: launch_arith_op<T, OP_BOR, false>(ctx, a, aligned); break;
          case OP_BXOR: launch_arith_op<T, OP_BXOR, false>(ctx, a, aligned); break;
          case OP_SHL: launch_arith_op<T, OP_SHL, false>(ctx, a, aligned); break;
          case OP_SHR: launch_arith_op<T, OP_SHR, false>(ctx, a, aligned); break;
          case OP_BANDNOT: launch_arith_op<T, OP_BANDNOT, false>(ctx, a, aligned); break;
          default: launch_arith_op<T, OP_BNOT, false>(ctx, a, aligned); break;
        }
      }
      break;
  }
}

ah_status dispatch_type(ah_context* ctx, ah_type t, int op, const ArithArgs& a, bool aligned) {
  switch (t) {
    case AH_INT8: launch_arith<int8_t>(ctx, op, a, aligned); break;
    case AH_INT16: launch_arith<int16_t>(ctx, op, a, aligned); break;
    case AH_INT32: launch_arith<int32_t>(ctx, op, a, aligned); break;
    case AH_INT64: launch_arith<int64_t>(ctx, op, a, aligned); break;
    case AH_UINT8: launch_arith<uint8_t>(ctx, op, a, aligned); break;
    case AH_UINT16: launch_arith<uint16_t>(ctx, op, a, aligned); break;
    case AH_UINT32: launch_arith<uint32_t>(ctx, op, a, aligned); break;
    case AH_UINT64: launch_arith<uint64_t>(ctx, op, a, aligned); break;
    case AH_FLOAT16: launch_arith<ah_f16>(ctx, op, a, aligned); break;
    case AH_FLOAT32: launch_arith<float>(ctx, op, a, aligned); break;
    case AH_FLOAT64: launch_arith<double>(ctx, op, a, aligned); break;
    default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported arithmetic type");
  }
  return AH_OK;
}

bool is_float_type(ah_type t) { return ah_type_is_float(t) || t == AH_FLOAT16; }

bool op_checked_for(ah_type t, int op) {
  if (is_float_type(t)) return false;
  return op == OP_ADD || op == OP_SUB || op == OP_MUL || op == OP_DIV || op == OP_REM || op == OP_NEG;
}

// read one element as i128-ish text for the error message ({:?} of a Rust int)
ah_status read_elem_text(ah_context* ctx, ah_type t, const void* base, int64_t idx, char* buf,
                         size_t n, bool* is_zero) {
  int w = ah_type_width(t);
  uint64_t raw = 0;
  AH_HIP(ctx, hipMemcpyAsync(&raw, (const char*)base + idx * w, w, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  *is_zero = raw == 0;
  switch (t) {
    case AH_INT8: snprintf(buf, n, "%d", (int)(int8_t)raw); break;
    case AH_INT16: snprintf(buf, n, "%d", (int)(int16_t)raw); break;
    case AH_INT32: snprintf(buf, n, "%d", (int)(int32_t)raw); break;
    case AH_INT64: snprintf(buf, n, "%lld", (long long)(int64_t)raw); break;
    default: snprintf(buf, n, "%llu", (unsigned long long)raw); break;
  }
  return AH_OK;
}

void free_out_bufs(ah_context* ctx, void* ov, size_t vbytes, void* ob, size_t bbytes) {
  ah_out_free(ctx, ov, vbytes);
  ah_out_free(ctx, ob, bbytes);
}

}  // namespace

extern "C" ah_status ah_arith_binary(ah_context* ctx, ah_arith_op op, const ah_array_view* lhs,
                                     int32_t l_s, const ah_array_view* rhs, int32_t r_s,
                                     ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !lhs || !rhs || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_ADD || op > AH_BIT_AND_NOT) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown arithmetic op %d", op);
  const bool bitwise = op >= AH_BIT_AND;
  if (bitwise) op = OP_BAND + (op - AH_BIT_AND);  // public 8..13 -> internal OP_BAND..OP_BANDNOT
  const ah_type t = lhs->type;
  // arithmetic_op (numeric.rs:225-275): both sides must be the same numeric type; bitwise.rs: integers
  if (lhs->type != rhs->type || !(ah_type_is_integer(t) || (is_float_type(t) && !bitwise)))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid arithmetic operation: %s %s %s",
                   ah_type_name(lhs->type), op_sym(op), ah_type_name(rhs->type));
  const bool checked = op_checked_for(t, op);
  const int w = ah_type_width(t);
  out->type = t;
  l_s = l_s != 0;
  r_s = r_s != 0;

  int64_t len;
  BitView va{nullptr, 0}, vb{nullptr, 0};
  bool want_valid = false, all_null = false;
  if (l_s == r_s) {
    // binary (arity.rs:115-123) / try_binary (:264-271)
    if (lhs->length != rhs->length)
      return ah_fail(ctx, AH_COMPUTE_ERROR,
                     checked ? "Cannot perform a binary operation on arrays of different length"
                             : "Cannot perform binary operation on arrays of different length");
    len = lhs->length;
    if (len == 0) return AH_OK;
    bool lp = lhs->validity != nullptr, rp = rhs->validity != nullptr;
    if (checked) {  // is_nullable(): null_count != 0 (arity.rs:279)
      int64_t ln = 0, rn = 0;
      AH_TRY(ah_resolve_null_count(ctx, lhs, &ln));
      AH_TRY(ah_resolve_null_count(ctx, rhs, &rn));
      want_valid = (ln != 0 || rn != 0);
    } else {
      want_valid = lp || rp;  // NullBuffer::union is presence-based (null.rs:79-88)
    }
    if (want_valid) {
      if (lp) va = make_bitview(lhs->validity, lhs->validity_bit_offset);
      if (rp) vb = make_bitview(rhs->validity, rhs->validity_bit_offset);
    }
  } else {
    // op!/try_op! with one scalar side (numeric.rs:278-317)
    const ah_array_view* arr = l_s ? rhs : lhs;
    const ah_array_view* sc = l_s ? lhs : rhs;
    len = arr->length;
    int64_t sn = 0;
    AH_TRY(ah_resolve_null_count(ctx, sc, &sn));
    if (sc->length < 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar datum must have length 1");
    if (sn != 0) all_null = true;  // PrimitiveArray::new_null(len)
    else if (arr->validity) {
      want_valid = true;  // nulls cloned (unary / try_unary)
      va = make_bitview(arr->validity, arr->validity_bit_offset);
    }
  }

  size_t vbytes = (size_t)len * w, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  if (all_null) {
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st != AH_OK) {
      ah_out_free(ctx, ov, vbytes);
      return st;
    }
    hipMemsetAsync(ov, 0, vbytes, ctx->stream);
    hipMemsetAsync(ob, 0, bbytes, ctx->stream);
    AH_HIP(ctx, ah_end_of_call_sync(ctx));
    out->length = len;
    out->values = ov;
    out->values_bytes = (int64_t)vbytes;
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len;
    return AH_OK;
  }
  int64_t set_bits = len;
  // The validity union (+ its popcount read-back) is one host wait.  Checked ops need the union BEFORE the value
  // kernel (it only evaluates valid slots); unchecked ops do not, so their value kernel is enqueued first and the
  // union's read-back is the call's ONE wait (it used to be two, with the GPU idle between them).
  const bool union_first = checked;
  if (want_valid) {
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st == AH_OK && union_first)
      st = ah_bitmap_op(ctx, (va.words && vb.words) ? BM_AND : BM_COPY, va.words ? va : vb, vb,
                        BitView{nullptr, 0}, len, (unsigned long long*)ob, &set_bits);
    if (st != AH_OK) {
      free_out_bufs(ctx, ov, vbytes, ob, bbytes);
      return st;
    }
  }
  unsigned long long* first_err = nullptr;
  if (checked) {
    ah_status st = ah_pool_alloc(ctx, 8, (void**)&first_err);
    if (st != AH_OK) {
      free_out_bufs(ctx, ov, vbytes, ob, bbytes);
      return st;
    }
    hipMemsetAsync(first_err, 0xFF, 8, ctx->stream);
  }
  ArithArgs a{};
  a.l = lhs->values;
  a.r = rhs->values;
  a.out = ov;
  a.len = len;
  a.l_scalar = (l_s != r_s) && l_s;
  a.r_scalar = (l_s != r_s) && r_s;
  a.valid = checked ? (const unsigned long long*)ob : nullptr;
  a.first_err = first_err;
  bool aligned = ((((uintptr_t)a.l) | ((uintptr_t)a.r) | ((uintptr_t)ov)) & 15) == 0;
  if (a.l_scalar) aligned = ((((uintptr_t)a.r) | ((uintptr_t)ov)) & 15) == 0;
  if (a.r_scalar) aligned = ((((uintptr_t)a.l) | ((uintptr_t)ov)) & 15) == 0;
  // small unchecked call: validity union, its popcount and the mailbox post ride in the value kernel
  const int64_t nblocks = ah_ceil_div(ah_ceil_div(len, aligned ? 16 / w : 1), 256 * 4);
  const bool fused_post = !checked && !ctx->deferred && want_valid && nblocks <= 2048;
  if (fused_post) {
    a.post = 1;
    a.va = va.words ? va : vb;  // (one side only: it is `va`)
    a.vb = (va.words && vb.words) ? vb : BitView{nullptr, 0};
    a.vout = (unsigned long long*)ob;
    a.total = ctx->scratch + AH_TICKET_COUNT;
  }
  ah_status st;
  {
    ah_prof_scope ps(ctx, "arith_binary");
    st = dispatch_type(ctx, t, op, a, aligned);
  }
  hipError_t e = hipGetLastError();
  bool waited = false;
  if (fused_post && st == AH_OK && e == hipSuccess) {
    e = ah_count_read(ctx, &set_bits);
    waited = true;
  } else if (fused_post) {
    ah_count_reset(ctx);  // the kernel may have been enqueued with its counting tail: never leave the counters dirty
  } else if (st == AH_OK && e == hipSuccess && want_valid && !union_first) {
    int64_t* cnt = AH_COUNT(ctx, &set_bits);
    st = ah_bitmap_op(ctx, (va.words && vb.words) ? BM_AND : BM_COPY, va.words ? va : vb, vb, BitView{nullptr, 0}, len,
                      (unsigned long long*)ob, cnt);
    waited = st == AH_OK && cnt != nullptr;  // the popcount read-back waited for the whole stream
  }
  if (st == AH_OK && e == hipSuccess && checked)
    e = ah_d2h(ctx, ctx->pinned, first_err, 8);
  // checked ops report device-side errors, so they stay synchronous even in deferred mode
  if (e == hipSuccess && !waited) e = checked ? ah_stream_wait(ctx) : ah_end_of_call_sync(ctx);
  ah_pool_free(ctx, first_err);
  if (st != AH_OK || e != hipSuccess) {
    free_out_bufs(ctx, ov, vbytes, ob, bbytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "arithmetic kernel failed: %s", hipGetErrorString(e));
  }
  if (checked && ctx->pinned[0] != ~0ull) {
    int64_t pos = (int64_t)ctx->pinned[0];
    free_out_bufs(ctx, ov, vbytes, ob, bbytes);
    char lt[32], rt[32];
    bool lz, rz;
    AH_TRY(read_elem_text(ctx, t, lhs->values, a.l_scalar ? 0 : pos, lt, sizeof lt, &lz));
    AH_TRY(read_elem_text(ctx, t, rhs->values, a.r_scalar ? 0 : pos, rt, sizeof rt, &rz));
    if ((op == AH_DIV || op == AH_REM) && rz) return ah_fail(ctx, AH_DIVIDE_BY_ZERO, "Divide by zero error");
    return ah_fail(ctx, AH_ARITHMETIC_OVERFLOW, "Overflow happened on: %s %s %s", lt, op_sym(op), rt);
  }
  out->length = len;
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (want_valid) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = checked ? len - set_bits : ah_nulls(ctx, len, set_bits);
  }
  return AH_OK;
}

static ah_status arith_unary(ah_context* ctx, const ah_array_view* v, int op, ah_array_out* out);

extern "C" ah_status ah_arith_neg(ah_context* ctx, const ah_array_view* v, int32_t wrapping,
                                  ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  const ah_type t = v->type;
  // neg (numeric.rs:103-178): signed ints (checked) and floats; neg_wrapping
  // (:181-186): every integer and float
  bool ok = is_float_type(t) || ah_type_is_signed(t) || (wrapping && ah_type_is_integer(t));
  if (!ok)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid arithmetic operation: !%s", ah_type_name(t));
  return arith_unary(ctx, v, wrapping ? OP_NEG_W : OP_NEG, out);
}

// bitwise_not (arrow-arith/src/bitwise.rs:113-120): `unary`, nulls cloned
extern "C" ah_status ah_bitwise_not(ah_context* ctx, const ah_array_view* v, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (!ah_type_is_integer(v->type))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid arithmetic operation: !%s", ah_type_name(v->type));
  return arith_unary(ctx, v, OP_BNOT, out);
}

static ah_status arith_unary(ah_context* ctx, const ah_array_view* v, int op, ah_array_out* out) {
  const ah_type t = v->type;
  const bool checked = op_checked_for(t, op);
  const int w = ah_type_width(t);
  const int64_t len = v->length;
  out->type = t;
  out->length = len;
  if (len == 0) return AH_OK;
  size_t vbytes = (size_t)len * w, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  int64_t set_bits = len;
  if (v->validity) {
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st == AH_OK)
      st = ah_bitmap_op(ctx, BM_COPY, make_bitview(v->validity, v->validity_bit_offset),
                        BitView{nullptr, 0}, BitView{nullptr, 0}, len, (unsigned long long*)ob,
                        checked ? &set_bits : AH_COUNT(ctx, &set_bits));
    if (st != AH_OK) {
      free_out_bufs(ctx, ov, vbytes, ob, bbytes);
      return st;
    }
  }
  unsigned long long* first_err = nullptr;
  if (checked) {
    ah_status st = ah_pool_alloc(ctx, 8, (void**)&first_err);
    if (st != AH_OK) {
      free_out_bufs(ctx, ov, vbytes, ob, bbytes);
      return st;
    }
    hipMemsetAsync(first_err, 0xFF, 8, ctx->stream);
  }
  ArithArgs a{};
  a.l = v->values;
  a.r = v->values;
  a.out = ov;
  a.len = len;
  a.valid = checked ? (const unsigned long long*)ob : nullptr;
  a.first_err = first_err;
  bool aligned = ((((uintptr_t)a.l) | ((uintptr_t)ov)) & 15) == 0;
  ah_status st = dispatch_type(ctx, t, op, a, aligned);
  hipError_t e = hipGetLastError();
  if (st == AH_OK && e == hipSuccess && checked)
    e = ah_d2h(ctx, ctx->pinned, first_err, 8);
  if (e == hipSuccess) e = checked ? ah_stream_wait(ctx) : ah_end_of_call_sync(ctx);
  ah_pool_free(ctx, first_err);
  if (st != AH_OK || e != hipSuccess) {
    free_out_bufs(ctx, ov, vbytes, ob, bbytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "neg kernel failed: %s", hipGetErrorString(e));
  }
  if (checked && ctx->pinned[0] != ~0ull) {
    int64_t pos = (int64_t)ctx->pinned[0];
    free_out_bufs(ctx, ov, vbytes, ob, bbytes);
    char lt[32];
    bool lz;
    AH_TRY(read_elem_text(ctx, t, v->values, pos, lt, sizeof lt, &lz));
    return ah_fail(ctx, AH_ARITHMETIC_OVERFLOW, "Overflow happened on: - %s", lt);
  }
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (v->validity) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = checked ? len - set_bits : ah_nulls(ctx, len, set_bits);
  }
  return AH_OK;
}
