// arith_decimal.hip — arrow_arith::numeric::{add, sub, mul, div, rem} on Decimal128: `decimal_op`
// (arrow-arith/src/numeric.rs:971-1103).
//
// The reference aligns the two scales with powers of ten and evaluates
//     l.mul_checked(l_mul)? <op>_checked (r.mul_checked(r_mul)?)
// per valid row over i128 (try_op!: union of the nulls, a null scalar gives an all-null result, the first failing
// row is the error); the result TYPE follows the Hive rules (add / sub: scale max(s1, s2), precision
// max(s1, s2) + max(p1 - s1, p2 - s2) + 1; mul: p1 + p2 + 1, s1 + s2; div: scale s1 + 4; rem: scale max(s1, s2)).
//
// MI355X design: lane-per-row over 16-byte natives — one dwordx4 load per operand and one dwordx4 store per row, the
// widest access a lane can issue, 48 B/row + validity, HBM-bound for add / sub / mul (a 128-bit checked multiply is
// four 64 x 64 -> 128 products on quarter-rate multipliers); div / rem run a shift-subtract long division whose trip
// count is the quotient's bit length, compute-bound by construction (the device runtime has no 128-bit division).
// The arithmetic and the type rules live in decimal_arith.hpp, verified on the host against native __int128
// arithmetic (tests/cpp/decimal_arith_host_test.cpp) before they run here.
#include "common.hpp"
#include "decimal_arith.hpp"

#include <algorithm>

namespace {

using namespace da;

struct alignas(16) Raw16 {
  uint64_t lo, hi;
};
__device__ __forceinline__ i128 to_i128(Raw16 r) { return (i128)(((u128)r.hi << 64) | r.lo); }
__device__ __forceinline__ Raw16 from_i128(i128 v) { return Raw16{(uint64_t)(u128)v, (uint64_t)((u128)v >> 64)}; }

struct DecArgs {
  const Raw16* l;
  const Raw16* r;
  Raw16* out;
  int64_t len;
  int l_scalar, r_scalar;
  const unsigned long long* valid;  // union words (offset 0) or nullptr = all valid
  unsigned long long* first_err;
  DParams p;
};

__global__ void __launch_bounds__(256) dec_kernel(DecArgs a) {
  const Raw16 ls = a.l_scalar ? a.l[0] : Raw16{0, 0};
  const Raw16 rs = a.r_scalar ? a.r[0] : Raw16{0, 0};
  unsigned long long err = ~0ull;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.len; i += 2 * stride) {
    // two rows in flight per lane: both operand pairs are requested before either is computed
    const int64_t j = i + stride;
    Raw16 l0 = a.l_scalar ? ls : a.l[i], r0 = a.r_scalar ? rs : a.r[i];
    Raw16 l1 = Raw16{0, 0}, r1 = Raw16{0, 0};
    if (j < a.len) {
      l1 = a.l_scalar ? ls : a.l[j];
      r1 = a.r_scalar ? rs : a.r[j];
    }
    bool v0 = true, v1 = j < a.len;
    if (a.valid) {
      v0 = (a.valid[i >> 6] >> (i & 63)) & 1;
      if (j < a.len) v1 = (a.valid[j >> 6] >> (j & 63)) & 1;
    }
    i128 o0 = 0, o1 = 0, t0, t1;
    if (v0 && dec_row(a.p, to_i128(l0), to_i128(r0), &o0, &t0, &t1) != D_OK) {
      o0 = 0;
      err = (unsigned long long)i < err ? (unsigned long long)i : err;
    }
    a.out[i] = from_i128(o0);
    if (j < a.len) {
      if (v1 && dec_row(a.p, to_i128(l1), to_i128(r1), &o1, &t0, &t1) != D_OK) {
        o1 = 0;
        err = (unsigned long long)j < err ? (unsigned long long)j : err;
      }
      a.out[j] = from_i128(o1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(err, o, 64);
    err = other < err ? other : err;
  }
  if ((threadIdx.x & 63) == 0 && err != ~0ull) atomicMin(a.first_err, err);
}

ah_status fetch_i128(ah_context* ctx, const void* base, int64_t idx, i128* out) {
  uint64_t raw[2] = {0, 0};
  AH_HIP(ctx, hipMemcpyAsync(raw, (const char*)base + idx * 16, 16, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  *out = (i128)(((u128)raw[1] << 64) | raw[0]);
  return AH_OK;
}

}  // namespace

// called by ah_arith_with_types (arith_temporal.hip) when both operands are AH_DT_DECIMAL128
ah_status ah_decimal_arith(ah_context* ctx, ah_arith_op op, const char* op_sym, const ah_array_view* lhs, int32_t l_s,
                           const ah_data_type* lt, const ah_array_view* rhs, int32_t r_s, const ah_data_type* rt,
                           ah_array_out* out, ah_data_type* out_type) {
  if (lhs->type != AH_FIXED16 || rhs->type != AH_FIXED16)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Decimal128 operands are 16-byte values (AH_FIXED16)");
  DPlan plan = make_decimal_plan(op, *lt, *rt, op_sym);
  if (plan.status != AH_OK) return ah_fail(ctx, plan.status, "%s", plan.message.c_str());
  *out_type = plan.result;
  out->type = AH_FIXED16;
  l_s = l_s != 0;
  r_s = r_s != 0;

  // try_op! (numeric.rs:296-317) over try_binary (arity.rs:254-309) / try_unary: the null rules of arith.hip's checked ops
  int64_t len;
  BitView va{nullptr, 0}, vb{nullptr, 0};
  bool want_valid = false, all_null = false;
  if (l_s == r_s) {
    if (lhs->length != rhs->length)
      return ah_fail(ctx, AH_COMPUTE_ERROR, "Cannot perform a binary operation on arrays of different length");
    len = lhs->length;
    int64_t ln = 0, rn = 0;
    AH_TRY(ah_resolve_null_count(ctx, lhs, &ln));
    AH_TRY(ah_resolve_null_count(ctx, rhs, &rn));
    want_valid = (ln != 0 || rn != 0);  // is_nullable(): null_count != 0
    if (want_valid) {
      if (lhs->validity) va = make_bitview(lhs->validity, lhs->validity_bit_offset);
      if (rhs->validity) vb = make_bitview(rhs->validity, rhs->validity_bit_offset);
    }
  } else {
    const ah_array_view* arr = l_s ? rhs : lhs;
    const ah_array_view* sc = l_s ? lhs : rhs;
    len = arr->length;
    if (sc->length < 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar datum must have length 1");
    int64_t sn = 0;
    AH_TRY(ah_resolve_null_count(ctx, sc, &sn));
    if (sn != 0) all_null = true;  // PrimitiveArray::new_null(len)
    else if (arr->validity) {
      want_valid = true;  // try_unary clones the nulls
      va = make_bitview(arr->validity, arr->validity_bit_offset);
    }
  }
  if (len == 0) {
    if (plan.post_status != AH_OK) return ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
    return AH_OK;
  }
  const size_t vbytes = (size_t)len * 16, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  auto release = [&] {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
  };
  int64_t set_bits = len;
  if (all_null || want_valid) {
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st != AH_OK) return release(), st;
  }
  if (all_null) {
    hipMemsetAsync(ov, 0, vbytes, ctx->stream);
    hipMemsetAsync(ob, 0, bbytes, ctx->stream);
    hipError_t e = ah_stream_wait(ctx);
    if (e != hipSuccess) return release(), ah_fail(ctx, AH_HIP_ERROR, "decimal arithmetic: %s", hipGetErrorString(e));
    set_bits = 0;
  } else {
    if (want_valid) {
      ah_status st = ah_bitmap_op(ctx, (va.words && vb.words) ? BM_AND : BM_COPY, va.words ? va : vb, vb, BitView{nullptr, 0},
                                  len, (unsigned long long*)ob, &set_bits);
      if (st != AH_OK) return release(), st;
    }
    unsigned long long* first_err = nullptr;
    ah_status st = ah_pool_alloc(ctx, 8, (void**)&first_err);
    if (st != AH_OK) return release(), st;
    hipMemsetAsync(first_err, 0xFF, 8, ctx->stream);
    DecArgs a{};
    a.l = (const Raw16*)lhs->values;
    a.r = (const Raw16*)rhs->values;
    a.out = (Raw16*)ov;
    a.len = len;
    a.l_scalar = (l_s != r_s) && l_s;
    a.r_scalar = (l_s != r_s) && r_s;
    a.valid = (const unsigned long long*)ob;
    a.first_err = first_err;
    a.p = plan.p;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(len, 512), 256 * 32));
    {
      ah_prof_scope ps(ctx, "arith_decimal");
      dec_kernel<<<grid, 256, 0, ctx->stream>>>(a);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, first_err, 8);
    ah_pool_free(ctx, first_err);
    if (e != hipSuccess) return release(), ah_fail(ctx, AH_HIP_ERROR, "decimal arithmetic kernel failed: %s", hipGetErrorString(e));
    if (ctx->pinned[0] != ~0ull) {
      const int64_t pos = (int64_t)ctx->pinned[0];
      release();
      i128 l = 0, r = 0, o = 0, sl = 0, sr = 0;
      AH_TRY(fetch_i128(ctx, lhs->values, a.l_scalar ? 0 : pos, &l));
      AH_TRY(fetch_i128(ctx, rhs->values, a.r_scalar ? 0 : pos, &r));
      const int fail = dec_row(plan.p, l, r, &o, &sl, &sr);  // the same closure on the host names the failing step
      std::string msg;
      ah_status es = row_error(plan.p, fail, l, r, sl, sr, &msg);
      return ah_fail(ctx, es, "%s", msg.c_str());
    }
  }
  if (plan.post_status != AH_OK) return release(), ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
  out->length = len;
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (ob) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len - set_bits;
  }
  return AH_OK;
}

// ------------------------------------------------------------------ Decimal128 -> Decimal128 cast
// cast_decimal_to_decimal_same_type (arrow-cast/src/cast/decimal.rs:448-489).  One launch, 16 B in / 16 B out per row;
// the three array-level shapes of apply_decimal_cast (:351-377): `unary` when every input fits (all slots, nulls
// cloned), `unary_opt` in safe mode (failures become nulls, a null buffer is always attached), `try_unary` otherwise
// (nulls cloned, the first failing valid row is the error).
namespace {

struct DCastArgs {
  const Raw16* in;
  Raw16* out;
  BitView in_valid;
  int64_t len;
  unsigned long long* out_valid;
  unsigned long long* block_valid;
  unsigned long long* first_err;
  int all_slots, fail_is_null;
  CParams p;
};

__global__ void __launch_bounds__(256) dcast_kernel(DCastArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long nvalid = 0, err = ~0ull;
  for (int64_t base = (int64_t)blockIdx.x * 512; base < a.len; base += (int64_t)gridDim.x * 512) {
    const int64_t wbase = base + wave * 128;
    Raw16 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int64_t i = wbase + k * 64 + lane;
      v[k] = i < a.len ? a.in[i] : Raw16{0, 0};
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t i0 = wbase + k * 64;
      if (i0 >= a.len) break;
      const int64_t i = i0 + lane;
      const uint64_t iv = bv_fetch64(a.in_valid, i0, a.len);
      const bool valid = (iv >> lane) & 1;
      i128 o = 0;
      int stage = 0;
      bool ok = true;
      if (valid || (a.all_slots && i < a.len)) ok = dcast_row(a.p, to_i128(v[k]), &o, &stage);
      if (!ok) {
        o = 0;
        if (!a.fail_is_null) err = (unsigned long long)i < err ? (unsigned long long)i : err;
      }
      if (i < a.len) a.out[i] = from_i128(o);
      if (a.out_valid) {
        const uint64_t w = a.fail_is_null ? (iv & __ballot(ok)) : iv;
        if (lane == 0) {
          a.out_valid[i0 >> 6] = w;
          nvalid += __popcll(w);
        }
      }
    }
  }
  if (!a.fail_is_null) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long other = __shfl_xor(err, o, 64);
      err = other < err ? other : err;
    }
    if (lane == 0 && err != ~0ull) atomicMin(a.first_err, err);
  }
  if (a.out_valid) {
    __shared__ unsigned long long s[4];
    if (lane == 0) s[wave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) a.block_valid[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
  }
}

__global__ void __launch_bounds__(256) dcast_sum_kernel(const unsigned long long* in, int64_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc += in[i];
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = s[0] + s[1] + s[2] + s[3];
}

}  // namespace

// called by ah_cast_with_types (cast_temporal.hip) when both types are AH_DT_DECIMAL128
ah_status ah_decimal_cast(ah_context* ctx, const ah_array_view* values, const ah_data_type* from, const ah_data_type* to,
                          int32_t safe, ah_array_out* out) {
  ah_out_init(out);
  if (values->type != AH_FIXED16) return ah_fail(ctx, AH_INVALID_ARGUMENT, "Decimal128 values are 16-byte (AH_FIXED16)");
  CPlan plan = make_decimal_cast_plan(*from, *to);
  if (plan.status != AH_OK) return ah_fail(ctx, plan.status, "%s", plan.message.c_str());
  const int64_t len = values->length;
  out->type = AH_FIXED16;
  out->length = len;
  const bool unary = plan.p.mode == C_CLONE || plan.p.infallible;
  const bool fail_is_null = !unary && safe;
  const bool want_valid = fail_is_null || values->validity != nullptr;
  if (len == 0) {
    if (plan.post_status != AH_OK) return ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
    return AH_OK;
  }
  const size_t vbytes = (size_t)len * 16, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  unsigned long long* aux = nullptr;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(len, 512), 256 * 16));
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  auto release = [&] {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    ah_out_init(out);
  };
  ah_status st = AH_OK;
  if (want_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st == AH_OK) st = ah_pool_alloc(ctx, (size_t)(grid + 4) * 8, (void**)&aux);
  if (st != AH_OK) return release(), st;
  hipMemsetAsync(aux, 0xFF, 8, ctx->stream);
  DCastArgs a{};
  a.in = (const Raw16*)values->values;
  a.out = (Raw16*)ov;
  a.in_valid = values->validity ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  a.len = len;
  a.out_valid = (unsigned long long*)ob;
  a.block_valid = aux + 2;
  a.first_err = aux;
  a.all_slots = unary;
  a.fail_is_null = (fail_is_null || unary) ? 1 : 0;
  a.p = plan.p;
  {
    ah_prof_scope ps(ctx, "cast_decimal");
    dcast_kernel<<<grid, 256, 0, ctx->stream>>>(a);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && want_valid) dcast_sum_kernel<<<1, 256, 0, ctx->stream>>>(a.block_valid, grid, aux + 1);
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, aux, 16);
  ah_pool_free(ctx, aux);
  if (e != hipSuccess) return release(), ah_fail(ctx, AH_HIP_ERROR, "decimal cast kernel failed: %s", hipGetErrorString(e));
  if (!a.fail_is_null && ctx->pinned[0] != ~0ull) {
    const int64_t pos = (int64_t)ctx->pinned[0];
    release();
    i128 x = 0;
    AH_TRY(fetch_i128(ctx, values->values, pos, &x));
    std::string msg;
    ah_status es = dcast_row_error(plan.p, *to, x, &msg);
    return ah_fail(ctx, es, "%s", msg.c_str());
  }
  if (plan.post_status != AH_OK) return release(), ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (want_valid) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len - (int64_t)ctx->pinned[1];
  }
  return AH_OK;
}

// ------------------------------------------------------------------ integer -> Decimal128 cast
// cast_integer_to_decimal (arrow-cast/src/cast/mod.rs:366-443): 1..8 B in, 16 B out per row; safe = unary_opt (a null
// buffer is always attached), unsafe = try_unary; a scale factor that does not exist in the source type = `unary` zeros.
namespace {

template <typename I>
__global__ void __launch_bounds__(256) icast_kernel(const I* in, Raw16* out, BitView in_valid, int64_t len,
                                                    unsigned long long* out_valid, unsigned long long* block_valid,
                                                    unsigned long long* first_err, int all_slots, int fail_is_null, IParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long nvalid = 0, err = ~0ull;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < len; base += (int64_t)gridDim.x * 1024) {
    const int64_t wbase = base + wave * 256;
    I v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i = wbase + k * 64 + lane;
      v[k] = i < len ? in[i] : I{};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i0 = wbase + k * 64;
      if (i0 >= len) break;
      const int64_t i = i0 + lane;
      const uint64_t iv = bv_fetch64(in_valid, i0, len);
      const bool valid = (iv >> lane) & 1;
      i128 o = 0;
      int stage = 0;
      bool ok = true;
      if (valid || (all_slots && i < len)) ok = icast_row(p, (i128)v[k], &o, &stage);
      if (!ok) {
        o = 0;
        if (!fail_is_null) err = (unsigned long long)i < err ? (unsigned long long)i : err;
      }
      if (i < len) out[i] = from_i128(o);
      if (out_valid) {
        const uint64_t w = fail_is_null ? (iv & __ballot(ok)) : iv;
        if (lane == 0) {
          out_valid[i0 >> 6] = w;
          nvalid += __popcll(w);
        }
      }
    }
  }
  if (!fail_is_null) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long other = __shfl_xor(err, o, 64);
      err = other < err ? other : err;
    }
    if (lane == 0 && err != ~0ull) atomicMin(first_err, err);
  }
  if (out_valid) {
    __shared__ unsigned long long s[4];
    if (lane == 0) s[wave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) block_valid[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
  }
}

}  // namespace

// called by ah_cast_with_types when `from` is a plain integer type and `to` is AH_DT_DECIMAL128
ah_status ah_int_to_decimal_cast(ah_context* ctx, const ah_array_view* values, const ah_data_type* to, int32_t safe,
                                 ah_array_out* out) {
  ah_out_init(out);
  const ah_type t = values->type;
  if (!ah_type_is_integer(t)) return ah_fail(ctx, AH_INVALID_ARGUMENT, "integer -> Decimal128 needs integer values");
  const int w = ah_type_width(t);
  const bool sg = ah_type_is_signed(t);
  const u128 src_max = sg ? (((u128)1 << (8 * w - 1)) - 1) : (((u128)1 << (8 * w)) - 1);
  IPlan plan = make_int_to_decimal_plan(src_max, *to);
  if (plan.status != AH_OK) return ah_fail(ctx, plan.status, "%s", plan.message.c_str());
  const int64_t len = values->length;
  out->type = AH_FIXED16;
  out->length = len;
  const bool unary = plan.p.mode == I_ZEROS;
  const bool fail_is_null = !unary && safe;
  const bool want_valid = fail_is_null || values->validity != nullptr;
  if (len == 0) {
    if (plan.post_status != AH_OK) return ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
    return AH_OK;
  }
  const size_t vbytes = (size_t)len * 16, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  unsigned long long* aux = nullptr;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(len, 1024), 256 * 16));
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  auto release = [&] {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    ah_out_init(out);
  };
  ah_status st = AH_OK;
  if (want_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st == AH_OK) st = ah_pool_alloc(ctx, (size_t)(grid + 4) * 8, (void**)&aux);
  if (st != AH_OK) return release(), st;
  hipMemsetAsync(aux, 0xFF, 8, ctx->stream);
  const BitView inv = values->validity ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  const int fin = (fail_is_null || unary) ? 1 : 0;
#define AH_ICAST(T) icast_kernel<T><<<grid, 256, 0, ctx->stream>>>((const T*)values->values, (Raw16*)ov, inv, len, (unsigned long long*)ob, aux + 2, aux, unary ? 1 : 0, fin, plan.p)
  {
    ah_prof_scope ps(ctx, "cast_int_decimal");
    switch (t) {
      case AH_INT8: AH_ICAST(int8_t); break;
      case AH_INT16: AH_ICAST(int16_t); break;
      case AH_INT32: AH_ICAST(int32_t); break;
      case AH_INT64: AH_ICAST(int64_t); break;
      case AH_UINT8: AH_ICAST(uint8_t); break;
      case AH_UINT16: AH_ICAST(uint16_t); break;
      case AH_UINT32: AH_ICAST(uint32_t); break;
      default: AH_ICAST(uint64_t); break;
    }
  }
#undef AH_ICAST
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && want_valid) dcast_sum_kernel<<<1, 256, 0, ctx->stream>>>(aux + 2, grid, aux + 1);
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, aux, 16);
  ah_pool_free(ctx, aux);
  if (e != hipSuccess) return release(), ah_fail(ctx, AH_HIP_ERROR, "integer -> decimal cast kernel failed: %s", hipGetErrorString(e));
  if (!fin && ctx->pinned[0] != ~0ull) {
    const int64_t pos = (int64_t)ctx->pinned[0];
    release();
    uint64_t raw = 0;
    AH_HIP(ctx, hipMemcpyAsync(&raw, (const char*)values->values + pos * w, w, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(ctx, ah_stream_wait(ctx));
    i128 x;
    if (sg) x = w == 1 ? (i128)(int8_t)raw : w == 2 ? (i128)(int16_t)raw : w == 4 ? (i128)(int32_t)raw : (i128)(int64_t)raw;
    else x = (i128)(u128)raw;
    std::string msg;
    ah_status es = icast_row_error(plan.p, *to, x, &msg);
    return ah_fail(ctx, es, "%s", msg.c_str());
  }
  if (plan.post_status != AH_OK) return release(), ah_fail(ctx, plan.post_status, "%s", plan.post_message.c_str());
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (want_valid) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len - (int64_t)ctx->pinned[1];
  }
  return AH_OK;
}
