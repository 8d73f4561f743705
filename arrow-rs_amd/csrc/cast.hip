// cast.hip — arrow_cast::cast on MI355X: numeric <-> numeric.
//
// Reference path: cast_with_options (arrow-cast/src/cast/mod.rs:790) numeric arms
// :1578-1697 -> cast_numeric_arrays :2550-2571 -> safe: numeric_cast :2606 ->
// PrimitiveArray::unary_opt (arrow-array/src/array/primitive_array.rs:1065-1102:
// valid slots only, null slots 0, failed conversions become null, ALWAYS a null
// buffer) / unsafe: try_numeric_cast :2575 -> try_unary (:990-1016: nulls cloned,
// first failure -> CastError "Can't cast value {v:?} to type {T}").
// num_cast == num_traits::cast (num-traits 0.2.19): int->float `as`; float->int
// succeeds iff trunc(v) is representable; int->int range-checked.
//
// MI355X design: lane-per-row, 4 independent rows in flight per lane; a wave's
// validity word is in_valid & __ballot(ok), so the output bitmap and null_count
// come for free.  String casts live in cast_string.hip.
#include "common.hpp"

#include <charconv>
#include <cmath>
#include <limits>
#include <type_traits>

ah_status ah_cast_to_string_view(ah_context* ctx, const ah_array_view* values, ah_array_out* out);  // cast_view.hip
ah_status ah_cast_to_string(ah_context* ctx, const ah_array_view* values, ah_type to_type,
                            ah_array_out* out);  // cast_string.hip
ah_status ah_cast_bool(ah_context* ctx, const ah_array_view* values, ah_type to_type, ah_array_out* out);  // cast_bool.hip
ah_status ah_cast_parse(ah_context* ctx, const ah_array_view* values, ah_type to_type, int32_t safe,
                        ah_array_out* out);  // cast_parse.hip
ah_status ah_cast_i64_via_f64_to_string(ah_context* ctx, const ah_array_view* values, ah_type to_type, ah_array_out* out);  // cast_string.hip

namespace {

template <typename I, typename O>
__device__ __forceinline__ bool num_cast(I v, O* out) {
  // Float16 on either side (half 2.7.1 num_traits.rs): `NumCast for f16` = n.to_f32().map(f16::from_f32), `ToPrimitive for f16`
  // = self.to_f32().to_X() — every conversion goes through f32 (i64 / f64 -> f16 round twice, as the reference does)
  if constexpr (std::is_same<I, ah_f16>::value && std::is_same<O, ah_f16>::value) {
    *out = v;
    return true;
  } else if constexpr (std::is_same<I, ah_f16>::value) {
    return num_cast<float, O>(ah_f16_to_f32(v), out);
  } else if constexpr (std::is_same<O, ah_f16>::value) {
    float f;
    num_cast<I, float>(v, &f);
    *out = ah_f32_to_f16(f);
    return true;
  } else if constexpr (std::is_same<I, O>::value) {
    *out = v;
    return true;
  } else if constexpr (std::is_floating_point<O>::value) {
    *out = (O)v;
    return true;
  } else if constexpr (std::is_floating_point<I>::value) {
    I t = trunc(v);  // NaN fails both comparisons
    constexpr int bits = sizeof(O) * 8 - (std::is_signed<O>::value ? 1 : 0);
    const I hi = (I)ldexp(1.0, bits);  // 2^bits, exact in f32/f64
    const I lo = std::is_signed<O>::value ? -hi : (I)0;
    if (!(t >= lo && t < hi)) return false;
    *out = (O)t;
    return true;
  } else {
    if constexpr (std::is_signed<I>::value && !std::is_signed<O>::value) {
      if (v < 0) return false;
      using UI = typename std::make_unsigned<I>::type;
      if (sizeof(I) > sizeof(O) && (UI)v > (UI)std::numeric_limits<O>::max()) return false;
    } else if constexpr (!std::is_signed<I>::value && std::is_signed<O>::value) {
      using UO = typename std::make_unsigned<O>::type;
      if (sizeof(I) >= sizeof(O) && v > (I)(UO)std::numeric_limits<O>::max()) return false;
    } else if constexpr (sizeof(I) > sizeof(O)) {
      if (v < (I)std::numeric_limits<O>::min() || v > (I)std::numeric_limits<O>::max()) return false;
    }
    *out = (O)v;
    return true;
  }
}

struct CastArgs {
  const void* in;
  void* out;
  BitView in_valid;  // words == nullptr: all valid
  int64_t len;
  unsigned long long* out_valid;    // nullptr: none
  unsigned long long* slots;        // 64 zero-state counters of valid output rows (ctx->scratch)
  unsigned long long* first_err;    // all-ones-state position word (unsafe mode)
  int safe;
};

// ---- general kernel: lane-per-row, any alignment (sliced inputs, 1/2-byte types)
template <typename I, typename O>
__global__ void __launch_bounds__(256) cast_kernel(CastArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const I* ip = (const I*)a.in;
  O* op = (O*)a.out;
  unsigned long long nvalid = 0, err = ~0ull;
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < a.len; base += (int64_t)gridDim.x * 1024) {
    const int64_t wbase = base + wave * 256;
    I v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i = wbase + k * 64 + lane;
      v[k] = i < a.len ? ip[i] : I{};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i0 = wbase + k * 64;
      if (i0 >= a.len) break;
      int64_t i = i0 + lane;
      uint64_t iv = bv_fetch64(a.in_valid, i0, a.len);
      bool valid = (iv >> lane) & 1;
      O o = O{};
      bool ok = true;
      if (valid) ok = num_cast<I, O>(v[k], &o);  // valid slots only; null slots stay 0
      if (!ok) {
        o = O{};
        if (!a.safe) {
          unsigned long long pos = (unsigned long long)i;
          err = pos < err ? pos : err;
        }
      }
      if (i < a.len) op[i] = o;
      if (a.out_valid) {
        uint64_t w = a.safe ? (iv & __ballot(ok)) : iv;
        if (lane == 0) {
          a.out_valid[i0 >> 6] = w;
          nvalid += __popcll(w);
        }
      }
    }
  }
  if (!a.safe) {
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
    if (threadIdx.x == 0) {
      unsigned long long c = s[0] + s[1] + s[2] + s[3];
      if (c) atomicAdd(&a.slots[blockIdx.x & 63], c);
    }
  }
}

// ---- streaming kernel: V consecutive rows per lane so that the wider side moves 16 bytes per lane per access
// (8 -> 8 bytes: both sides), short-lived workgroups with 4 x 16-byte loads in flight per lane — the shape that
// reaches 5.6 TB/s on a read+write stream on MI355X, against 4.7 TB/s for 8-byte accesses from a persistent
// grid.  A group of 64*V rows is V validity words: per element slot e the wave ballots "valid and converted",
// and lane k < V bit-interleaves its slices of the V ballots into output word k on the vector unit (as cmp.hip does).

template <typename T, int V> struct alignas(sizeof(T) * V) CastVec { T e[V]; };
#ifndef AH_CAST_G
#define AH_CAST_G 4
#endif
constexpr int CAST_G = AH_CAST_G;  // groups in flight per wave

template <typename I, typename O, int V>
__global__ void __launch_bounds__(256) cast_stream_kernel(CastArgs a) {
  using VI = CastVec<I, V>;
  using VO = CastVec<O, V>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const I* ip = (const I*)a.in;
  O* op = (O*)a.out;
  const int64_t ngroups = (a.len + 64 * V - 1) / (64 * V);
  const int64_t nwords = (a.len + 63) >> 6;
  const int64_t g0 = ((int64_t)blockIdx.x * 4 + ah_uniform(wave)) * CAST_G;
  unsigned long long nvalid = 0, err = ~0ull;
  VI iv[CAST_G];
  uint32_t vb[CAST_G];
#pragma unroll
  for (int gi = 0; gi < CAST_G; ++gi) {  // every load of the wave's groups goes out first
    const int64_t i = ((g0 + gi) * 64 + lane) * V;
    vb[gi] = 0;
    if (i + V <= a.len) {
      iv[gi] = ah_ld_stream<ah_nt_l(true)>((const VI*)(ip + i));
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) iv[gi].e[e] = (i + e < a.len) ? ip[i + e] : I{};
    }
    vb[gi] = bv_lane_bits<V>(a.in_valid, (g0 + gi) * 64 * V, lane, a.len);  // scalar loads: see common.hpp
  }
#pragma unroll
  for (int gi = 0; gi < CAST_G; ++gi) {
    const int64_t g = g0 + gi;
    if (g >= ngroups) break;
    const int64_t i = (g * 64 + lane) * V;
    VO ov;
    uint64_t ballots[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const bool valid = (vb[gi] >> e) & 1u;
      O o = O{};
      bool ok = true;
      if (valid) ok = num_cast<I, O>(iv[gi].e[e], &o);
      if (!ok) {
        o = O{};
        if (!a.safe && i + e < a.len) {
          unsigned long long pos = (unsigned long long)(i + e);
          err = pos < err ? pos : err;
        }
      }
      ov.e[e] = o;
      ballots[e] = __ballot(valid && (ok || !a.safe) && (i + e < a.len));
    }
    if (i + V <= a.len) {
      ah_st_stream<ah_nt_s(true)>((VO*)(op + i), ov);
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (i + e < a.len) op[i + e] = ov.e[e];
    }
    if (a.out_valid) {
      // word k of the group covers lanes [k*64/V, (k+1)*64/V): lane k < V interleaves its slices of the V ballots (vector unit)
      const int ksh = (lane & (V - 1)) * (64 / V);
      uint64_t mine = 0;
#pragma unroll
      for (int e = 0; e < V; ++e) mine |= vspread<V>(ballots[e] >> ksh) << e;
      const int64_t wi = g * V + lane;
      if (lane < V && wi < nwords) {
        a.out_valid[wi] = mine;
        nvalid += __popcll(mine);
      }
    }
  }
  if (!a.safe) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long other = __shfl_xor(err, o, 64);
      err = other < err ? other : err;
    }
    if (lane == 0 && err != ~0ull) atomicMin(a.first_err, err);
  }
  if (a.out_valid) {
    nvalid = wave_reduce_add64(nvalid);
    __shared__ unsigned long long s[4];
    if (lane == 0) s[wave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long c = s[0] + s[1] + s[2] + s[3];
      if (c) atomicAdd(&a.slots[blockIdx.x & 63], c);
    }
  }
}

// after the cast: mail[0] = first failing position (~0: none), mail[1] = valid output rows; both scratch
// areas go back to their rest state and the mailbox is posted (mail == nullptr in deferred mode: clean only)
__global__ void __launch_bounds__(64) cast_finish_kernel(unsigned long long* slots, unsigned long long* first_err,
                                                         uint64_t* mail, uint64_t seq) {
  unsigned long long v = slots[threadIdx.x];
  slots[threadIdx.x] = 0;
  v = wave_reduce_add64(v);
  if (threadIdx.x == 0) {
    const unsigned long long e = *first_err;
    *first_err = ~0ull;
    if (mail) {
      __hip_atomic_store(mail, (uint64_t)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(mail + 1, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ah_mail_post(mail, seq);
    }
  }
}

template <typename I, typename O>
void launch_cast(ah_context* ctx, const CastArgs& a) {
  constexpr int WMAX = sizeof(I) > sizeof(O) ? sizeof(I) : sizeof(O);
  constexpr int V = WMAX >= 4 ? 16 / WMAX : 4;
  const bool aligned = (((uintptr_t)a.in) % (sizeof(I) * V) == 0) && (((uintptr_t)a.out) % (sizeof(O) * V) == 0);
  if (aligned) {
    const int64_t ngroups = ah_ceil_div(a.len, 64 * (int64_t)V);
    const int64_t grid = std::max<int64_t>(1, ah_ceil_div(ngroups, 4 * CAST_G));
    cast_stream_kernel<I, O, V><<<(unsigned)grid, 256, 0, ctx->stream>>>(a);
  } else {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(a.len, 1024), 256 * 16));
    cast_kernel<I, O><<<grid, 256, 0, ctx->stream>>>(a);
  }
}

template <typename I>
ah_status launch_from(ah_context* ctx, ah_type to, const CastArgs& a) {
  switch (to) {
    case AH_INT8: launch_cast<I, int8_t>(ctx, a); break;
    case AH_INT16: launch_cast<I, int16_t>(ctx, a); break;
    case AH_INT32: launch_cast<I, int32_t>(ctx, a); break;
    case AH_INT64: launch_cast<I, int64_t>(ctx, a); break;
    case AH_UINT8: launch_cast<I, uint8_t>(ctx, a); break;
    case AH_UINT16: launch_cast<I, uint16_t>(ctx, a); break;
    case AH_UINT32: launch_cast<I, uint32_t>(ctx, a); break;
    case AH_UINT64: launch_cast<I, uint64_t>(ctx, a); break;
    case AH_FLOAT16: launch_cast<I, ah_f16>(ctx, a); break;
    case AH_FLOAT32: launch_cast<I, float>(ctx, a); break;
    case AH_FLOAT64: launch_cast<I, double>(ctx, a); break;
    default: return ah_fail(ctx, AH_CAST_ERROR, "unsupported cast target");
  }
  return AH_OK;
}

bool is_numeric(ah_type t) { return ah_type_is_integer(t) || ah_type_is_float(t); }
bool is_numeric16(ah_type t) { return is_numeric(t) || t == AH_FLOAT16; }  // the numeric <-> numeric arms (cast/mod.rs:1578-1697) take Float16 too

// Rust `{:?}` of a float (core::fmt float_to_general_debug): shortest digits,
// exponential iff |v| >= 1e16 or 0 < |v| < 1e-4.  Host-side, error text only.
template <typename F>
std::string rust_debug_float(F v) {
  if (v != v) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  std::string s;
  if (std::signbit(v)) s += '-';
  if (v == 0) return s + "0.0";
  char sci[64];
  auto res = std::to_chars(sci, sci + sizeof sci, std::fabs(v), std::chars_format::scientific);
  *res.ptr = 0;
  std::string digits;
  const char* p = sci;
  for (; *p && *p != 'e'; ++p) if (*p != '.') digits += *p;
  int exp10 = atoi(p + 1);
  int nd = (int)digits.size();
  int kk = exp10 + 1;
  if (kk >= 17 || kk <= -4) {
    s += digits[0];
    if (nd > 1) s += "." + digits.substr(1);
    return s + "e" + std::to_string(kk - 1);
  }
  if (kk <= 0) return s + "0." + std::string((size_t)-kk, '0') + digits;
  if (nd <= kk) return s + digits + std::string((size_t)(kk - nd), '0') + ".0";
  return s + digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
}

ah_status elem_debug_text(ah_context* ctx, ah_type t, const void* base, int64_t idx, std::string* out) {
  int w = ah_type_width(t);
  uint64_t raw = 0;
  AH_HIP(ctx, hipMemcpyAsync(&raw, (const char*)base + idx * w, w, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  char buf[40];
  switch (t) {
    case AH_INT8: snprintf(buf, sizeof buf, "%d", (int)(int8_t)raw); break;
    case AH_INT16: snprintf(buf, sizeof buf, "%d", (int)(int16_t)raw); break;
    case AH_INT32: snprintf(buf, sizeof buf, "%d", (int)(int32_t)raw); break;
    case AH_INT64: snprintf(buf, sizeof buf, "%lld", (long long)(int64_t)raw); break;
    case AH_FLOAT16: {  // Debug for f16 prints its f32 value
      ah_f16 h;
      h.bits = (uint16_t)raw;
      *out = rust_debug_float<float>(ah_f16_to_f32(h));
      return AH_OK;
    }
    case AH_FLOAT32: {
      float f;
      memcpy(&f, &raw, 4);
      *out = rust_debug_float<float>(f);
      return AH_OK;
    }
    case AH_FLOAT64: {
      double d;
      memcpy(&d, &raw, 8);
      *out = rust_debug_float<double>(d);
      return AH_OK;
    }
    default: snprintf(buf, sizeof buf, "%llu", (unsigned long long)raw); break;
  }
  *out = buf;
  return AH_OK;
}

}  // namespace

extern "C" int32_t ah_can_cast_types(ah_type from, ah_type to) {
  if (from == to) return ah_type_width(from) >= 0 || from == AH_UTF8 || from == AH_LARGE_UTF8;
  if (is_numeric16(from) && is_numeric16(to)) return 1;
  if ((from == AH_BOOL && is_numeric(to)) || (is_numeric(from) && to == AH_BOOL)) return 1;  // cast/mod.rs:254-255
  if (is_numeric(from) && (to == AH_UTF8 || to == AH_LARGE_UTF8)) return 1;
  if ((from == AH_UTF8 || from == AH_LARGE_UTF8) && is_numeric(to)) return 1;  // cast/mod.rs `(Utf8, _)` parse arms
  // cast/mod.rs:278 `(Utf8 | LargeUtf8, Utf8View)`, :281 `(_, Utf8View) => from_type.is_primitive()`
  if (to == AH_UTF8_VIEW && (is_numeric(from) || from == AH_UTF8 || from == AH_LARGE_UTF8)) return 1;
  return 0;
}

extern "C" ah_status ah_cast(ah_context* ctx, const ah_array_view* values, ah_type to_type,
                             int32_t safe, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  const ah_type from = values->type;
  if (!ah_can_cast_types(from, to_type))
    return ah_fail(ctx, AH_CAST_ERROR, "Casting from %s to %s not supported", ah_type_name(from),
                   ah_type_name(to_type));
  if (to_type == AH_UTF8_VIEW) return ah_cast_to_string_view(ctx, values, out);  // cast/mod.rs:1546 (numbers), :1302 / :1432 (Utf8 / LargeUtf8)
  if ((from == AH_UTF8 || from == AH_LARGE_UTF8) && is_numeric(to_type)) return ah_cast_parse(ctx, values, to_type, safe, out);
  if (to_type == AH_UTF8 || to_type == AH_LARGE_UTF8) return ah_cast_to_string(ctx, values, to_type, out);
  if ((from == AH_BOOL) != (to_type == AH_BOOL)) return ah_cast_bool(ctx, values, to_type, out);
  const int64_t len = values->length;
  const int wo = ah_type_width(to_type);
  out->type = to_type;
  out->length = len;
  const size_t vbytes = (size_t)len * wo, bbytes = ah_bitmap_bytes(len);

  if (from == to_type) {  // cast_with_options :797-799 — a clone
    if (len == 0) return AH_OK;
    void* ov = nullptr;
    AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
    hipMemcpyAsync(ov, values->values, vbytes, hipMemcpyDeviceToDevice, ctx->stream);
    out->values = ov;
    out->values_bytes = (int64_t)vbytes;
    if (values->validity) {
      void* ob = nullptr;
      int64_t set = 0;
      ah_status st = ah_out_alloc(ctx, bbytes, &ob);
      if (st == AH_OK)
        st = ah_bitmap_op(ctx, BM_COPY, make_bitview(values->validity, values->validity_bit_offset),
                          BitView{nullptr, 0}, BitView{nullptr, 0}, len, (unsigned long long*)ob,
                          AH_COUNT(ctx, &set));
      if (st != AH_OK) {
        ah_out_free(ctx, ov, vbytes);
        ah_out_free(ctx, ob, bbytes);
        ah_out_init(out);
        return st;
      }
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = ah_nulls(ctx, len, set);
    }
    AH_HIP(ctx, ah_end_of_call_sync(ctx));
    return AH_OK;
  }

  // safe: unary_opt ALWAYS carries a null buffer (primitive_array.rs:1098-1102);
  // unsafe: try_unary clones the input nulls (presence-based)
  const bool want_valid = safe || values->validity != nullptr;
  if (len == 0) {
    // unary_opt on an empty array still yields Some(empty NullBuffer); nothing to allocate
    return AH_OK;
  }
  void* ov = nullptr;
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  ah_status st = AH_OK;
  if (want_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st != AH_OK) {
    ah_out_free(ctx, ov, vbytes);
    return st;
  }
  CastArgs a{};
  a.in = values->values;
  a.out = ov;
  a.in_valid = values->validity ? make_bitview(values->validity, values->validity_bit_offset)
                                : BitView{nullptr, 0};
  a.len = len;
  a.out_valid = (unsigned long long*)ob;
  a.slots = ctx->scratch;                       // zero between calls
  a.first_err = ctx->scratch + AH_SCRATCH_ONES;  // all-ones between calls
  a.safe = safe ? 1 : 0;
  {
    ah_prof_scope ps(ctx, "cast_numeric");
    switch (from) {
      case AH_INT8: st = launch_from<int8_t>(ctx, to_type, a); break;
      case AH_INT16: st = launch_from<int16_t>(ctx, to_type, a); break;
      case AH_INT32: st = launch_from<int32_t>(ctx, to_type, a); break;
      case AH_INT64: st = launch_from<int64_t>(ctx, to_type, a); break;
      case AH_UINT8: st = launch_from<uint8_t>(ctx, to_type, a); break;
      case AH_UINT16: st = launch_from<uint16_t>(ctx, to_type, a); break;
      case AH_UINT32: st = launch_from<uint32_t>(ctx, to_type, a); break;
      case AH_UINT64: st = launch_from<uint64_t>(ctx, to_type, a); break;
      case AH_FLOAT16: st = launch_from<ah_f16>(ctx, to_type, a); break;
      case AH_FLOAT32: st = launch_from<float>(ctx, to_type, a); break;
      default: st = launch_from<double>(ctx, to_type, a); break;
    }
  }
  hipError_t e = hipGetLastError();
  if (st == AH_OK && e == hipSuccess) {
    // safe mode cannot fail, so in deferred mode it returns here with the null count unknown;
    // unsafe mode reports the first failing value and stays synchronous
    if (safe && ctx->deferred) {
      cast_finish_kernel<<<1, 64, 0, ctx->stream>>>(a.slots, a.first_err, nullptr, 0);
      out->values = ov;
      out->values_bytes = (int64_t)vbytes;
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = -1;
      return AH_OK;
    }
    const uint64_t seq = ah_mail_next(ctx);
    cast_finish_kernel<<<1, 64, 0, ctx->stream>>>(a.slots, a.first_err, ctx->pinned_dev, seq);
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  }
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "cast kernel failed: %s", hipGetErrorString(e));
  }
  if (!safe && ctx->pinned[0] != ~0ull) {
    int64_t pos = (int64_t)ctx->pinned[0];
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    std::string txt;
    AH_TRY(elem_debug_text(ctx, from, values->values, pos, &txt));
    return ah_fail(ctx, AH_CAST_ERROR, "Can't cast value %s to type %s", txt.c_str(), ah_type_name(to_type));
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

// A chain of casts as ONE call: cast(cast(values, types[0]), types[1]) ... — what BASELINE configs[3] runs (Int64 -> Float64 ->
// Utf8; arrow-cast/src/cast/mod.rs:1664 then :1549-1553 / cast/string.rs:21-39).  The cast analogue of ah_filter_expr: the caller
// says what it wants, the library decides what to materialise.  Int64 -> Float64 -> Utf8 / LargeUtf8 formats straight from the
// Int64 column (`v as f64` in registers): the Float64 intermediate — 4.3 GB written by X1 and read twice by X2 at 2^29 rows — is
// never built.  Every other chain runs its steps in sequence, releasing each intermediate as soon as the next step has it.
// Byte-exact against the step-by-step result (tests/test_gpu_parity.py::test_cast_chain_*, the 2^29-row test).
extern "C" ah_status ah_cast_chain(ah_context* ctx, const ah_array_view* values, int32_t n_types, const ah_type* types, int32_t safe,
                                   ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !out || n_types < 1 || !types) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  ah_type from = values->type;
  for (int i = 0; i < n_types; ++i) {  // the whole chain is validated before anything runs
    if (!ah_can_cast_types(from, types[i]))
      return ah_fail(ctx, AH_CAST_ERROR, "Casting from %s to %s not supported", ah_type_name(from), ah_type_name(types[i]));
    from = types[i];
  }
  static const char* fuse_env = getenv("AH_CAST_CHAIN_FUSE");  // "0": always step by step (A/B runs)
  if (n_types == 2 && values->type == AH_INT64 && types[0] == AH_FLOAT64 && (types[1] == AH_UTF8 || types[1] == AH_LARGE_UTF8) &&
      !(fuse_env && fuse_env[0] == '0'))
    return ah_cast_i64_via_f64_to_string(ctx, values, types[1], out);  // (Int64 -> Float64 cannot fail or null anything: safe is moot)
  ah_array_out cur;
  ah_out_init(&cur);
  ah_array_view v = *values;
  for (int i = 0; i < n_types; ++i) {
    ah_array_out next;
    const ah_status st = ah_cast(ctx, &v, types[i], safe, &next);
    if (i > 0) ah_array_release(ctx, &cur);  // (the pool reuses it in stream order, behind the step that read it)
    if (st != AH_OK) return st;
    cur = next;
    memset(&v, 0, sizeof v);
    v.type = cur.type;
    v.length = cur.length;
    v.null_count = cur.validity ? cur.null_count : 0;
    v.values = cur.values;
    v.values_bit_offset = cur.values_bit_offset;
    v.validity = cur.validity;
    v.validity_bit_offset = cur.validity_bit_offset;
    v.offsets = cur.offsets;
  }
  *out = cur;
  return AH_OK;
}
