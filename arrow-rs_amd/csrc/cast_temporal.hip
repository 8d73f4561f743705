// cast_temporal.hip — arrow_cast::cast_with_options, the "temporal casts" block
// (arrow-cast/src/cast/mod.rs:1700-2260): Date32 / Date64 / Time32 / Time64 / Timestamp / Duration
// among themselves and to / from the integers.
//
// The reference resolves every (from, to) pair to ONE elementwise closure over i32 / i64 handed to
// PrimitiveArray::unary (all slots, nulls cloned), unary_opt (valid slots, failures become null, a null
// buffer is always attached) or try_unary (valid slots, first failure is the error, nulls cloned)
// (arrow-array/src/array/primitive_array.rs:861,:1065,:990), sometimes followed by a second cast (Date ->
// Timestamp with a zone re-enters the Timestamp -> Timestamp arm).  Here a host-side planner turns the
// pair into at most two steps — a numeric ah_cast or one launch of `tcast_kernel` (8 or 12 B in, 4 or 8 B
// out per row: HBM-bound streaming like cast.hip, same lane-per-row layout, validity word =
// in_valid & __ballot(ok)) — so the calendar arithmetic (chrono 0.4.45, a third-party crate absent from
// /root/reference: floor division of seconds into days since 1970-01-01, NaiveDate::MIN..=MAX =
// -262143-01-01..=+262142-12-31) runs on the device next to the data.
#include "common.hpp"
#include "temporal_cast.hpp"

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

namespace {

using namespace tc;

struct TArgs {
  const void* in;
  void* out;
  BitView in_valid;  // words == nullptr: all valid
  int64_t len;
  unsigned long long* out_valid;    // nullptr: none
  unsigned long long* block_valid;  // per-block valid counts (when out_valid)
  unsigned long long* first_err;
  int all_slots;     // `unary`: the closure also runs on null slots
  int fail_is_null;  // unary_opt; otherwise a failure is reported through first_err
  TParams p;
};

template <typename I, typename O, int OP>
__global__ void __launch_bounds__(256) tcast_kernel(TArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const I* ip = (const I*)a.in;
  O* op = (O*)a.out;
  unsigned long long nvalid = 0, err = ~0ull;
  // The closures cost 100-300 SIMD cycles per 64 rows (64-bit multiplies are quarter rate), so the next tile's values
  // and validity words are requested BEFORE the current tile is computed: without this software prefetch a wave has
  // nothing in flight while it computes and compute time adds to memory time instead of hiding under it.
  const int64_t stride = (int64_t)gridDim.x * 1024;
  I v[4], nv[4];
  uint64_t iv[4], niv[4];
  auto load_tile = [&](int64_t base, I (&tv)[4], uint64_t (&tiv)[4]) {
    const int64_t wbase = base + wave * 256;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i = wbase + k * 64 + lane;
      tv[k] = i < a.len ? ip[i] : I{};
      tiv[k] = bv_fetch64(a.in_valid, wbase + k * 64, a.len);
    }
  };
  int64_t base = (int64_t)blockIdx.x * 1024;
  if (base < a.len) load_tile(base, v, iv);
  for (; base < a.len; base += stride) {
    const int64_t wbase = base + wave * 256;
    const bool more = base + stride < a.len;
    if (more) load_tile(base + stride, nv, niv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i0 = wbase + k * 64;
      if (i0 >= a.len) break;
      int64_t i = i0 + lane;
      bool valid = (iv[k] >> lane) & 1;
      O o = O{};
      bool ok = true;
      if (valid || (a.all_slots && i < a.len)) ok = tc_row_op<I, O, OP>(a.p, v[k], &o);
      if (!ok) {
        o = O{};
        if (!a.fail_is_null) {
          unsigned long long pos = (unsigned long long)i;
          err = pos < err ? pos : err;
        }
      }
      if (i < a.len) op[i] = o;
      if (a.out_valid) {
        uint64_t w = a.fail_is_null ? (iv[k] & __ballot(ok)) : iv[k];
        if (lane == 0) {
          a.out_valid[i0 >> 6] = w;
          nvalid += __popcll(w);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = nv[k];
        iv[k] = niv[k];
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

__global__ void __launch_bounds__(1024) tcast_sum_kernel(const unsigned long long* in, int64_t n,
                                                         unsigned long long* out) {
  unsigned long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += in[i];
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[16];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; i++) t += s[i];
    *out = t;
  }
}

ah_array_view view_with_nulls(const ah_array_out& o) {
  ah_array_view v{};
  v.type = o.type;
  v.length = o.length;
  v.null_count = o.validity ? o.null_count : 0;
  v.values = o.values;
  v.validity = o.validity;
  v.validity_bit_offset = o.validity_bit_offset;
  return v;
}

ah_status source_value_text(ah_context* ctx, ah_type t, const void* base, int64_t idx, std::string* out) {
  int w = ah_type_width(t);
  uint64_t raw = 0;
  AH_HIP(ctx, hipMemcpyAsync(&raw, (const char*)base + idx * w, w, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  *out = std::to_string(t == AH_INT32 ? (long long)(int32_t)raw : (long long)(int64_t)raw);
  return AH_OK;
}

// one instantiation per (arm, layout pair) the planner can produce
template <int OP>
bool launch_op(ah_context* ctx, ah_type from, ah_type to, const TArgs& a, int grid) {
  constexpr bool in32 = OP == T_MUL_WRAP || OP == T_MUL_CHECKED || OP == T_DIV;                  // Date32 / Time32 sources
  constexpr bool out32 = OP == T_MUL_CHECKED || OP == T_DIV || OP == T_DIV_TRY_I32 || OP == T_TS_DATE32 || OP == T_TS_TIME;
  constexpr bool out64 = OP != T_DIV_TRY_I32 && OP != T_TS_DATE32;
  if constexpr (in32 && out32 && OP != T_MUL_WRAP) {
    if (from == AH_INT32 && to == AH_INT32) return tcast_kernel<int32_t, int32_t, OP><<<grid, 256, 0, ctx->stream>>>(a), true;
  }
  if constexpr (in32 && out64 && OP != T_DIV) {
    if (from == AH_INT32 && to == AH_INT64) return tcast_kernel<int32_t, int64_t, OP><<<grid, 256, 0, ctx->stream>>>(a), true;
  }
  if constexpr (out32 && OP != T_MUL_CHECKED) {
    if (from == AH_INT64 && to == AH_INT32) return tcast_kernel<int64_t, int32_t, OP><<<grid, 256, 0, ctx->stream>>>(a), true;
  }
  if constexpr (out64) {
    if (from == AH_INT64 && to == AH_INT64) return tcast_kernel<int64_t, int64_t, OP><<<grid, 256, 0, ctx->stream>>>(a), true;
  }
  return false;  // a layout pair the planner never produces for this arm
}
bool launch_step(ah_context* ctx, ah_type from, ah_type to, const TArgs& a, int grid) {
  switch (a.p.op) {
    case T_MUL_WRAP: return launch_op<T_MUL_WRAP>(ctx, from, to, a, grid);
    case T_MUL_CHECKED: return launch_op<T_MUL_CHECKED>(ctx, from, to, a, grid);
    case T_DIV: return launch_op<T_DIV>(ctx, from, to, a, grid);
    case T_DIV_TRY_I32: return launch_op<T_DIV_TRY_I32>(ctx, from, to, a, grid);
    case T_TS_DATE32: return launch_op<T_TS_DATE32>(ctx, from, to, a, grid);
    case T_TS_TIME: return launch_op<T_TS_TIME>(ctx, from, to, a, grid);
    default: return launch_op<T_TZ_ADJUST>(ctx, from, to, a, grid);
  }
}

ah_status run_kernel_step(ah_context* ctx, const ah_array_view* values, const Step& step, int32_t safe,
                          ah_array_out* out) {
  ah_out_init(out);
  const ah_type from = values->type, to = step.to_phys;
  const int64_t len = values->length;
  out->type = to;
  out->length = len;
  const bool fail_is_null = step.mode == Step::OPT_OR_TRY && safe;
  // unary_opt ALWAYS carries a null buffer; unary / try_unary clone the input's (presence-based)
  const bool want_valid = fail_is_null || values->validity != nullptr;
  if (len == 0) return AH_OK;
  const size_t vbytes = (size_t)len * ah_type_width(to), bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  void* ob = nullptr;
  unsigned long long* aux = nullptr;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(len, 1024), 256 * 16));
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  ah_status st = AH_OK;
  if (want_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st == AH_OK) st = ah_pool_alloc(ctx, (size_t)(grid + 4) * 8, (void**)&aux);
  if (st != AH_OK) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    return st;
  }
  hipMemsetAsync(aux, 0xFF, 8, ctx->stream);
  TArgs a{};
  a.p = step.a;
  a.in = values->values;
  a.out = ov;
  a.in_valid = values->validity ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  a.len = len;
  a.out_valid = (unsigned long long*)ob;
  a.block_valid = aux + 2;
  a.first_err = aux;
  a.all_slots = step.mode == Step::UNARY;
  a.fail_is_null = (fail_is_null || step.mode == Step::UNARY) ? 1 : 0;  // UNARY steps cannot fail
  bool launched = false;
  {
    ah_prof_scope ps(ctx, "cast_temporal");
    launched = launch_step(ctx, from, to, a, grid);
  }
  hipError_t e = launched ? hipGetLastError() : hipErrorInvalidValue;
  const bool can_fail = step.mode != Step::UNARY && !fail_is_null;
  if (e == hipSuccess) {
    // infallible shapes run deferred like the safe numeric casts (cast.hip)
    if (!can_fail && ctx->deferred) {
      ah_pool_free(ctx, aux);
      out->values = ov;
      out->values_bytes = (int64_t)vbytes;
      out->validity = (uint8_t*)ob;
      out->validity_bytes = want_valid ? (int64_t)bbytes : 0;
      out->null_count = want_valid ? -1 : 0;
      return AH_OK;
    }
    if (want_valid) tcast_sum_kernel<<<1, 1024, 0, ctx->stream>>>(a.block_valid, grid, aux + 1);
    e = ah_d2h_wait(ctx, ctx->pinned, aux, 16);
  }
  ah_pool_free(ctx, aux);
  if (e != hipSuccess) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    ah_out_init(out);
    return ah_fail(ctx, AH_HIP_ERROR, "temporal cast kernel failed: %s", hipGetErrorString(e));
  }
  if (can_fail && ctx->pinned[0] != ~0ull) {
    int64_t pos = (int64_t)ctx->pinned[0];
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    ah_out_init(out);
    std::string txt;
    AH_TRY(source_value_text(ctx, from, values->values, pos, &txt));
    return ah_fail(ctx, step.err_status, step.err_fmt.c_str(), txt.c_str());
  }
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (want_valid) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len - (int64_t)ctx->pinned[1];
  }
  return AH_OK;
}

ah_status not_supported(ah_context* ctx, const ah_data_type& f, const ah_data_type& t) {
  return ah_fail(ctx, AH_CAST_ERROR, "Casting from %s to %s not supported", type_text(f).c_str(), type_text(t).c_str());
}

}  // namespace

ah_status ah_decimal_cast(ah_context* ctx, const ah_array_view* values, const ah_data_type* from, const ah_data_type* to,
                          int32_t safe, ah_array_out* out);  // arith_decimal.hip

ah_status ah_int_to_decimal_cast(ah_context* ctx, const ah_array_view* values, const ah_data_type* to, int32_t safe,
                                 ah_array_out* out);  // arith_decimal.hip
static bool plain_integer(int32_t id) { return id >= AH_INT8 && id <= AH_UINT64; }

static bool cast_not_built(int32_t id) { return id == AH_DT_DECIMAL128 || id == AH_DT_INTERVAL; }

extern "C" int32_t ah_can_cast_data_types(const ah_data_type* from, const ah_data_type* to) {
  if (!from || !to) return 0;
  if (from->id == AH_DT_DECIMAL128 && to->id == AH_DT_DECIMAL128) return 1;  // cast/mod.rs:178-181
  if (plain_integer(from->id) && to->id == AH_DT_DECIMAL128) return 1;          // :183-192
  if (cast_not_built(from->id) || cast_not_built(to->id)) return 0;
  if (!is_temporal(from->id) && !is_temporal(to->id)) return ah_can_cast_types((ah_type)from->id, (ah_type)to->id);
  std::vector<Step> plan;
  return make_plan(*from, *to, &plan) ? 1 : 0;
}

extern "C" ah_status ah_cast_with_types(ah_context* ctx, const ah_array_view* values, const ah_data_type* from,
                                        const ah_data_type* to, int32_t safe, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !from || !to || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (from->id == AH_DT_DECIMAL128 && to->id == AH_DT_DECIMAL128) return ah_decimal_cast(ctx, values, from, to, safe, out);
  if (plain_integer(from->id) && to->id == AH_DT_DECIMAL128) {
    if (values->type != (ah_type)from->id) return ah_fail(ctx, AH_INVALID_ARGUMENT, "values do not have the layout `from` names");
    return ah_int_to_decimal_cast(ctx, values, to, safe, out);
  }
  if (cast_not_built(from->id) || cast_not_built(to->id))
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "decimal <-> non-decimal and interval casts are not built on the device");
  if (!is_temporal(from->id) && !is_temporal(to->id)) {
    if (values->type != (ah_type)from->id)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "values are %s but `from` says %s", ah_type_name(values->type), type_text(*from).c_str());
    return ah_cast(ctx, values, (ah_type)to->id, safe, out);
  }
  if (values->type != physical_of(*from))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "values are %s but %s is stored as %s", ah_type_name(values->type),
                   type_text(*from).c_str(), ah_type_name(physical_of(*from)));
  std::vector<Step> plan;
  if (!make_plan(*from, *to, &plan)) return not_supported(ctx, *from, *to);
  ah_array_out cur;
  ah_out_init(&cur);
  ah_array_view v = *values;
  for (size_t i = 0; i < plan.size(); ++i) {
    ah_array_out next;
    ah_status st = plan[i].kind == Step::NUMERIC ? ah_cast(ctx, &v, plan[i].to_phys, safe, &next)
                                                 : run_kernel_step(ctx, &v, plan[i], safe, &next);
    if (i > 0) ah_array_release(ctx, &cur);
    if (st != AH_OK) return st;
    cur = next;
    v = view_with_nulls(cur);
  }
  *out = cur;
  return AH_OK;
}
