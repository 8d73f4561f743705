// cast_parse.hip — arrow_cast::cast Utf8 / LargeUtf8 -> integer and float types on MI355X: the inverse of
// cast_string.hip's Float64 -> Utf8 (SURVEY.md §8 rows X1/X2 run backwards), so text columns become numbers in HBM.
//
// Reference: cast_with_options (arrow-cast/src/cast/mod.rs:790) arms `(Utf8 | LargeUtf8, <numeric>)` ->
// parse_string::<P, O> (arrow-cast/src/cast/string.rs:66-74) -> parse_string_iter :87-120:
//   safe:   PrimitiveArray::from_trusted_len_iter(iter.map(|x| x.and_then(P::parse))) — unparsable text becomes
//           null, null / failed slots hold 0, and a null buffer is ALWAYS attached;
//   unsafe: the first unparsable valid row is CastError("Cannot cast string '{v}' to value of {P::DATA_TYPE} type"),
//           otherwise the input's null buffer is cloned.
// `P::parse` is restated in parse_num.hpp (parse.rs:446-528 + atoi 3.1.0 + lexical-core 1.0.6).
//
// MI355X design: a wave owns 64 consecutive rows (one validity word); a lane pulls its row (<= 32 bytes) into four
// registers with unaligned 8-byte loads and parses from registers, so the only memory traffic is offsets + text
// read once and values + 1 bit written once.  The validity word is `in_valid & __ballot(parsed)`.  Float rows
// whose 19-digit truncation straddles a rounding boundary are flagged in a per-row bitmap (one more ballot) and
// finished by parse_slow_kernel (exact big-integer comparison in scratch memory) — launched only if any exist.
#include "common.hpp"
#include "parse_num.hpp"

#include <algorithm>
#include <string>
#include <type_traits>

namespace {

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
// up to 8 bytes at p, never touching memory at or past `end`
__device__ __forceinline__ uint64_t load_tail(const uint8_t* p, const uint8_t* end) {
  if (p + 8 <= end) return load_u64_unaligned(p);
  uint64_t v = 0;
  for (int k = 0; p + k < end; ++k) v |= (uint64_t)p[k] << (8 * k);
  return v;
}
struct MemBytes {
  const uint8_t* p;
  __device__ __forceinline__ uint8_t operator[](int64_t i) const { return p[i]; }
};
struct RegBytes {
  uint64_t r0, r1, r2, r3;
  __device__ __forceinline__ uint8_t operator[](int64_t i) const {
    const uint64_t w = i < 16 ? (i < 8 ? r0 : r1) : (i < 24 ? r2 : r3);
    return (uint8_t)(w >> ((i & 7) * 8));
  }
};
struct Reg2Bytes {  // rows of at most 16 bytes (most numeric text): one select per byte instead of three
  uint64_t r0, r1;
  __device__ __forceinline__ uint8_t operator[](int64_t i) const {
    return (uint8_t)((i < 8 ? r0 : r1) >> ((i & 7) * 8));
  }
};

template <typename T> struct OutBits { using type = T; };
template <> struct OutBits<float> { using type = uint32_t; };
template <> struct OutBits<double> { using type = uint64_t; };

// one row -> (ok, value bits, needs the slow path)
template <typename T, typename B>
__device__ __forceinline__ bool parse_row(const B& s, int64_t n, typename OutBits<T>::type* out, bool* slow) {
  *slow = false;
  if constexpr (std::is_floating_point<T>::value) {
    PnDec d;
    int64_t b, e;
    if (!pn_scan_float(s, n, &d, &b, &e)) return false;
    *out = pn_convert<T>(d, slow);
    return true;
  } else {
    T v;
    if (!pn_parse_int<T>(s, n, &v)) return false;
    *out = v;
    return true;
  }
}

struct ParseArgs {
  const void* offs;
  const uint8_t* data;
  BitView in_valid;  // words == nullptr: all valid
  int64_t len;
  void* out;
  unsigned long long* out_valid;  // safe: in_valid & parsed; unsafe: copy of in_valid (nullptr: none)
  unsigned long long* slow_bits;  // floats only
  unsigned long long* counters;   // [0..63] valid-slot counts, [64] first failing row (unsafe), [65] slow rows
  int safe;
};

// KU words (64 rows each) per wave iteration: all offset loads, then all text loads, are issued before any parsing,
// so a wave keeps KU x 2 dependent round trips in flight instead of one (the loop is latency-bound otherwise).
template <typename OFF, typename T, int KU>
__global__ __launch_bounds__(256) void parse_kernel(ParseArgs a) {
  using U = typename OutBits<T>::type;
  const OFF* offs = (const OFF*)a.offs;
  U* out = (U*)a.out;
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (a.len + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const uint8_t* data_end = a.data + (int64_t)offs[a.len];
  unsigned long long nvalid = 0, nslow = 0, err = ~0ull;
  for (int64_t w0 = wave0 * KU; w0 < nwords; w0 += nwaves * KU) {
    int64_t a0[KU], n[KU];
    unsigned long long vw[KU];
    bool live[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int64_t row = (w0 + k) * 64 + lane;
      vw[k] = bv_fetch64(a.in_valid, (w0 + k) * 64, a.len);
      live[k] = row < a.len && ((vw[k] >> lane) & 1);
      a0[k] = live[k] ? (int64_t)offs[row] : 0;
      n[k] = live[k] ? (int64_t)offs[row + 1] - a0[k] : 0;
    }
    uint64_t r0[KU], r1[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const uint8_t* sp = a.data + a0[k];
      r0[k] = n[k] > 0 ? load_tail(sp, data_end) : 0;
      r1[k] = n[k] > 8 ? load_tail(sp + 8, data_end) : 0;
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int64_t w = w0 + k;
      const bool have = w < nwords;  // wave-uniform
      const int64_t row = w * 64 + lane;
      U v = 0;
      bool ok = false, slow = false;
      if (live[k]) {
        const uint8_t* sp = a.data + a0[k];
        if (n[k] <= 16) {
          ok = parse_row<T>(Reg2Bytes{r0[k], r1[k]}, n[k], &v, &slow);
        } else if (n[k] <= 32) {
          RegBytes rb;
          rb.r0 = r0[k];
          rb.r1 = r1[k];
          rb.r2 = load_tail(sp + 16, data_end);
          rb.r3 = n[k] > 24 ? load_tail(sp + 24, data_end) : 0;
          ok = parse_row<T>(rb, n[k], &v, &slow);
        } else {
          ok = parse_row<T>(MemBytes{sp}, n[k], &v, &slow);
        }
        if (!ok) {
          v = 0;
          const unsigned long long pos = (unsigned long long)row;
          err = pos < err ? pos : err;
        }
      }
      if (row < a.len) out[row] = v;
      const unsigned long long okw = __ballot(ok);
      if (lane == 0 && have) {
        if (a.out_valid) a.out_valid[w] = a.safe ? okw : vw[k];
        nvalid += __popcll(okw);
      }
      if constexpr (std::is_floating_point<T>::value) {
        const unsigned long long sw = __ballot(slow);
        if (lane == 0 && have) {
          a.slow_bits[w] = sw;
          nslow += __popcll(sw);
        }
      }
    }
  }
  if (lane == 0 && nvalid) atomicAdd(&a.counters[wave0 & 63], nvalid);
  if (lane == 0 && nslow) atomicAdd(&a.counters[65], nslow);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(err, o, 64);
    err = other < err ? other : err;
  }
  if (!a.safe && lane == 0 && err != ~0ull) atomicMin(&a.counters[64], err);
}

// rows flagged by parse_kernel: scan again from memory, decide the rounding exactly
template <typename OFF, typename F>
__global__ __launch_bounds__(64) void parse_slow_kernel(const OFF* offs, const uint8_t* data, int64_t len,
                                                        const unsigned long long* slow_bits, typename OutBits<F>::type* out) {
  const int lane = threadIdx.x;
  const int64_t nwords = (len + 63) >> 6;
  PnBig lhs, rhs;
  for (int64_t w = blockIdx.x; w < nwords; w += gridDim.x) {
    const unsigned long long sw = slow_bits[w];
    if (!sw) continue;
    if (!((sw >> lane) & 1)) continue;
    const int64_t row = w * 64 + lane;
    const int64_t a0 = (int64_t)offs[row], n = (int64_t)offs[row + 1] - a0;
    const MemBytes s{data + a0};
    PnDec d;
    int64_t b, e;
    if (!pn_scan_float(s, n, &d, &b, &e)) continue;  // cannot happen: the fast pass accepted it
    const PnBin lo = pn_compute_float<F>(d.q, d.w);
    typename OutBits<F>::type bits = pn_bits<F>(d.neg, lo);
    if (pn_slow_round_up<F>(s, b, e, lo, &lhs, &rhs)) bits = pn_next_up_magnitude<F>(bits);
    out[row] = bits;
  }
}

constexpr int PARSE_KU = 1;  // 2 measured slower (2.24 vs 2.07 ms per 2^27 Float64 rows): the loop is issue-bound, not latency-bound

template <typename OFF, typename T>
void launch_parse(ah_context* ctx, const ParseArgs& a, int grid) {
  parse_kernel<OFF, T, PARSE_KU><<<grid, 256, 0, ctx->stream>>>(a);
}

template <typename OFF>
ah_status launch_by_type(ah_context* ctx, ah_type to, const ParseArgs& a, int grid) {
  switch (to) {
    case AH_INT8: launch_parse<OFF, int8_t>(ctx, a, grid); break;
    case AH_INT16: launch_parse<OFF, int16_t>(ctx, a, grid); break;
    case AH_INT32: launch_parse<OFF, int32_t>(ctx, a, grid); break;
    case AH_INT64: launch_parse<OFF, int64_t>(ctx, a, grid); break;
    case AH_UINT8: launch_parse<OFF, uint8_t>(ctx, a, grid); break;
    case AH_UINT16: launch_parse<OFF, uint16_t>(ctx, a, grid); break;
    case AH_UINT32: launch_parse<OFF, uint32_t>(ctx, a, grid); break;
    case AH_UINT64: launch_parse<OFF, uint64_t>(ctx, a, grid); break;
    case AH_FLOAT32: launch_parse<OFF, float>(ctx, a, grid); break;
    case AH_FLOAT64: launch_parse<OFF, double>(ctx, a, grid); break;
    default: return ah_fail(ctx, AH_CAST_ERROR, "unsupported cast target");
  }
  return AH_OK;
}

}  // namespace

ah_status ah_cast_parse(ah_context* ctx, const ah_array_view* values, ah_type to_type, int32_t safe, ah_array_out* out) {
  const ah_type from = values->type;
  const bool large = from == AH_LARGE_UTF8;
  const int64_t len = values->length;
  const int wo = ah_type_width(to_type);
  out->type = to_type;
  out->length = len;
  if (len == 0) return AH_OK;
  if (!values->offsets) return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
  const bool is_float = ah_type_is_float(to_type);
  const bool want_valid = safe || values->validity != nullptr;
  const size_t vbytes = (size_t)len * wo, bbytes = ah_bitmap_bytes(len);
  void *ov = nullptr, *ob = nullptr, *slow = nullptr, *ctr = nullptr;
  auto cleanup = [&](bool outputs) {
    if (outputs) {
      ah_out_free(ctx, ov, vbytes);
      ah_out_free(ctx, ob, bbytes);
    }
    ah_pool_free(ctx, slow);
    ah_pool_free(ctx, ctr);
  };
  ah_status st = ah_out_alloc(ctx, vbytes, &ov);
  if (st == AH_OK && want_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st == AH_OK && is_float) st = ah_pool_alloc(ctx, bbytes, &slow);
  if (st == AH_OK) st = ah_pool_alloc(ctx, 66 * 8, &ctr);
  if (st != AH_OK) {
    cleanup(true);
    return st;
  }
  hipMemsetAsync(ctr, 0, 66 * 8, ctx->stream);
  hipMemsetAsync((unsigned long long*)ctr + 64, 0xFF, 8, ctx->stream);
  ParseArgs a{};
  a.offs = values->offsets;
  a.data = (const uint8_t*)values->values;
  a.in_valid = values->validity ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  a.len = len;
  a.out = ov;
  a.out_valid = (unsigned long long*)ob;
  a.slow_bits = (unsigned long long*)slow;
  a.counters = (unsigned long long*)ctr;
  a.safe = safe ? 1 : 0;
  const int64_t nwords = (len + 63) >> 6;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(256 * 32, ah_ceil_div(nwords, 4 * PARSE_KU)));
  {
    ah_prof_scope ps(ctx, "cast_parse");
    st = large ? launch_by_type<int64_t>(ctx, to_type, a, grid) : launch_by_type<int32_t>(ctx, to_type, a, grid);
  }
  hipError_t e = hipGetLastError();
  if (st == AH_OK && e == hipSuccess) e = ah_d2h(ctx, ctx->pinned, ctr, 66 * 8);
  if (st == AH_OK && e == hipSuccess) e = ah_stream_wait(ctx);
  if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in string -> numeric cast", hipGetErrorString(e));
  if (st != AH_OK) {
    cleanup(true);
    return st;
  }
  int64_t nvalid = 0;
  for (int i = 0; i < 64; ++i) nvalid += (int64_t)ctx->pinned[i];
  const uint64_t first_err = ctx->pinned[64];
  const int64_t nslow = (int64_t)ctx->pinned[65];
  if (!safe && first_err != ~0ull) {
    // CastError with the offending text (string.rs:108-113)
    const size_t ow = large ? 8 : 4;
    uint8_t raw[16];
    int64_t o0 = 0, o1 = 0;
    e = hipMemcpyAsync(raw, (const uint8_t*)values->offsets + first_err * ow, 2 * ow, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e == hipSuccess) {
      if (large) {
        memcpy(&o0, raw, 8);
        memcpy(&o1, raw + 8, 8);
      } else {
        int32_t t[2];
        memcpy(t, raw, 8);
        o0 = t[0];
        o1 = t[1];
      }
    }
    std::string text((size_t)(o1 - o0), '\0');
    if (e == hipSuccess && !text.empty()) {
      e = hipMemcpyAsync(&text[0], (const uint8_t*)values->values + o0, text.size(), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = ah_stream_wait(ctx);
    }
    cleanup(true);
    if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s reading the failing row", hipGetErrorString(e));
    return ah_fail(ctx, AH_CAST_ERROR, "Cannot cast string '%s' to value of %s type", text.c_str(), ah_type_name(to_type));
  }
  if (nslow > 0) {
    ah_prof_scope ps(ctx, "cast_parse_slow");
    const int sgrid = (int)std::max<int64_t>(1, std::min<int64_t>(nwords, 256 * 64));
    if (to_type == AH_FLOAT64) {
      if (large) parse_slow_kernel<int64_t, double><<<sgrid, 64, 0, ctx->stream>>>((const int64_t*)a.offs, a.data, len, a.slow_bits, (uint64_t*)ov);
      else parse_slow_kernel<int32_t, double><<<sgrid, 64, 0, ctx->stream>>>((const int32_t*)a.offs, a.data, len, a.slow_bits, (uint64_t*)ov);
    } else {
      if (large) parse_slow_kernel<int64_t, float><<<sgrid, 64, 0, ctx->stream>>>((const int64_t*)a.offs, a.data, len, a.slow_bits, (uint32_t*)ov);
      else parse_slow_kernel<int32_t, float><<<sgrid, 64, 0, ctx->stream>>>((const int32_t*)a.offs, a.data, len, a.slow_bits, (uint32_t*)ov);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) {
      cleanup(true);
      return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in the exact float parse", hipGetErrorString(e));
    }
  }
  cleanup(false);
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (want_valid) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    // unsafe mode clones the input nulls: every valid row parsed, so the count is the input's
    out->null_count = len - nvalid;
  }
  return AH_OK;
}
