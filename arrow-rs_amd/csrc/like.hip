// like.hip — SQL LIKE-family predicates on Utf8 / LargeUtf8 columns against a scalar pattern: the string side of
// "predicate construction" (SURVEY.md §8f row 2), so `like(col, "%ar%") -> filter(col, …)` stays in HBM.
//
// Reference: arrow_string::like::{like, nlike, starts_with, ends_with, contains} (arrow-string/src/like.rs:83-205)
// -> like_op (:218) -> op_scalar (:349) -> Predicate::{like, contains, StartsWith, EndsWith}
// (arrow-string/src/predicate.rs:44-120).  The reference picks Eq / StartsWith / EndsWith / Contains when the
// pattern allows and otherwise compiles a regex (regex_like, predicate.rs:246-306: `%` -> `.*`, `_` -> `.` with
// dot-matches-newline, `\x` -> literal x, a trailing `\` -> literal backslash, everything else literal,
// anchored at both ends).  All of those are the same language: literal bytes, `_` = one UTF-8 character,
// `%` = any run of characters.  MI355X design: the host tokenises the pattern once into that three-symbol
// program, one lane matches one row with the classic two-pointer wildcard walk (backtracking to the last `%`,
// whole characters at a time so `_` never starts inside a multi-byte character), the result word is a ballot.
// BooleanArray::from_unary semantics: values are computed for every slot, the input's null buffer is cloned.
#include "common.hpp"

#include <string>
#include <vector>

namespace {

enum : uint16_t { TOK_ANY1 = 256, TOK_STAR = 257 };  // 0..255 = that literal byte

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

__device__ __forceinline__ int utf8_len(uint8_t lead) {
  return lead < 0x80 ? 1 : (lead < 0xE0 ? 2 : (lead < 0xF0 ? 3 : 4));
}

// the row's bytes: either a pointer into HBM or, for rows of at most 32 bytes, four registers (no memory access
// inside the matching loop: the byte-at-a-time walk is a chain of dependent loads otherwise — 3.7 ms for `%99%`
// over 2^27 rows with global byte loads, 2.9 ms with the rows staged in LDS)
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

template <typename B>
__device__ __forceinline__ bool like_match(const B& s, int64_t n, const uint16_t* pat, int m) {
  int64_t i = 0, star_i = 0;
  int p = 0, star_p = -1;
  while (i < n) {
    const uint16_t t = p < m ? pat[p] : (uint16_t)0xFFFF;
    if (t < 256 && s[i] == (uint8_t)t) {
      ++i;
      ++p;
    } else if (t == TOK_ANY1) {
      i += utf8_len(s[i]);
      ++p;
    } else if (t == TOK_STAR) {
      star_p = p++;
      star_i = i;
    } else if (star_p >= 0) {  // let the last `%` swallow one more character and retry what follows it
      star_i += utf8_len(s[star_i]);
      i = star_i;
      p = star_p + 1;
    } else {
      return false;
    }
  }
  if (i > n) return false;  // a `_` ran past a truncated character (cannot happen on valid UTF-8)
  while (p < m && pat[p] == TOK_STAR) ++p;
  return p == m;
}

// up to 8 bytes at p, never touching memory at or past `end`
__device__ __forceinline__ uint64_t load_tail(const uint8_t* p, const uint8_t* end) {
  if (p + 8 <= end) return load_u64_unaligned(p);
  uint64_t v = 0;
  for (int k = 0; p + k < end; ++k) v |= (uint64_t)p[k] << (8 * k);
  return v;
}

template <typename O>
__global__ __launch_bounds__(256) void like_kernel(const O* offs, const uint8_t* data, int64_t len, const uint16_t* pat,
                                                   int m, int neg, unsigned long long* out) {
  __shared__ uint16_t s_pat[512];
  for (int i = threadIdx.x; i < m && i < 512; i += 256) s_pat[i] = pat[i];
  __syncthreads();
  const uint16_t* pp = m <= 512 ? s_pat : pat;
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (len + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const uint8_t* data_end = data + (int64_t)offs[len];
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t row = w * 64 + lane;
    bool res = false;
    if (row < len) {
      const int64_t a0 = (int64_t)offs[row], n = (int64_t)offs[row + 1] - a0;
      const uint8_t* sp = data + a0;
      if (n <= 16) {
        Reg2Bytes rb;
        rb.r0 = n > 0 ? load_tail(sp, data_end) : 0;
        rb.r1 = n > 8 ? load_tail(sp + 8, data_end) : 0;
        res = like_match(rb, n, pp, m);
      } else if (n <= 32) {
        RegBytes rb;
        rb.r0 = n > 0 ? load_tail(sp, data_end) : 0;
        rb.r1 = n > 8 ? load_tail(sp + 8, data_end) : 0;
        rb.r2 = n > 16 ? load_tail(sp + 16, data_end) : 0;
        rb.r3 = n > 24 ? load_tail(sp + 24, data_end) : 0;
        res = like_match(rb, n, pp, m);
      } else {
        res = like_match(MemBytes{sp}, n, pp, m);
      }
      res = res != (neg != 0);
    }
    const unsigned long long word = __ballot(res);
    if (lane == 0) out[w] = word;
  }
}

// LIKE pattern -> token program (regex_like, predicate.rs:246-306)
void tokenize_like(const std::string& pat, std::vector<uint16_t>* prog) {
  for (size_t i = 0; i < pat.size(); ++i) {
    const uint8_t c = (uint8_t)pat[i];
    if (c == '\\') {
      if (i + 1 < pat.size()) {  // the escaped CHARACTER is literal: copy all of its bytes
        const uint8_t lead = (uint8_t)pat[i + 1];
        const int n = lead < 0x80 ? 1 : (lead < 0xE0 ? 2 : (lead < 0xF0 ? 3 : 4));
        for (int k = 0; k < n && i + 1 + k < pat.size(); ++k) prog->push_back((uint8_t)pat[i + 1 + k]);
        i += (size_t)n;
      } else {
        prog->push_back('\\');  // trailing backslash: a literal backslash
      }
    } else if (c == '%') {
      if (prog->empty() || prog->back() != TOK_STAR) prog->push_back(TOK_STAR);
    } else if (c == '_') {
      prog->push_back(TOK_ANY1);
    } else {
      prog->push_back(c);
    }
  }
}

const char* op_name(int op) {
  switch (op) {
    case AH_LIKE: return "LIKE";
    case AH_NLIKE: return "NLIKE";
    case AH_STARTS_WITH: return "STARTS_WITH";
    case AH_ENDS_WITH: return "ENDS_WITH";
    default: return "CONTAINS";
  }
}

}  // namespace

extern "C" ah_status ah_string_like(ah_context* ctx, ah_like_op op, const ah_array_view* values,
                                    const ah_array_view* pattern, int32_t pattern_is_scalar, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !pattern || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_LIKE || op > AH_CONTAINS) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown string predicate %d", op);
  const ah_type t = values->type;
  // like_op (like.rs:236-293): both sides the same string type
  if (!(t == AH_UTF8 || t == AH_LARGE_UTF8) || pattern->type != t)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid string/binary operation: %s %s %s", ah_type_name(t), op_name(op),
                   ah_type_name(pattern->type));
  if (!pattern_is_scalar) {
    if (values->length != pattern->length)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "Cannot compare arrays of different lengths, got %lld vs %lld",
                     (long long)values->length, (long long)pattern->length);
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "string predicates with a per-row pattern array (only scalar patterns run on the device)");
  }
  if (pattern->length < 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar datum must have length 1");
  if (!values->offsets || !pattern->offsets) return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
  const int64_t len = values->length;
  out->type = AH_BOOL;
  out->length = len;
  if (len == 0) return AH_OK;
  const size_t bytes = ah_bitmap_bytes(len);
  int64_t pn = 0;
  AH_TRY(ah_resolve_null_count(ctx, pattern, &pn));
  unsigned long long* vals = nullptr;
  unsigned long long* nb = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, (void**)&vals));
  if (pn > 0) {  // null pattern: BooleanArray::new_null(len) (like.rs:314)
    ah_status st = ah_out_alloc(ctx, bytes, (void**)&nb);
    if (st != AH_OK) {
      ah_out_free(ctx, vals, bytes);
      return st;
    }
    hipMemsetAsync(vals, 0, bytes, ctx->stream);
    hipMemsetAsync(nb, 0, bytes, ctx->stream);
    hipError_t e = ah_stream_wait(ctx);
    if (e != hipSuccess) {
      ah_out_free(ctx, vals, bytes);
      ah_out_free(ctx, nb, bytes);
      return ah_fail(ctx, AH_HIP_ERROR, "string predicate failed: %s", hipGetErrorString(e));
    }
    out->values = vals;
    out->values_bytes = (int64_t)bytes;
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len;
    return AH_OK;
  }
  // the pattern is one short string: bring it to the host and compile it
  const size_t ow = t == AH_UTF8 ? 4 : 8;
  int64_t po[2] = {0, 0};
  {
    uint8_t raw[16];
    hipError_t e = hipMemcpyAsync(raw, pattern->offsets, 2 * ow, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) {
      ah_out_free(ctx, vals, bytes);
      return ah_fail(ctx, AH_HIP_ERROR, "reading the pattern failed: %s", hipGetErrorString(e));
    }
    if (ow == 4) {
      int32_t o32[2];
      memcpy(o32, raw, 8);
      po[0] = o32[0];
      po[1] = o32[1];
    } else {
      memcpy(po, raw, 16);
    }
  }
  std::string pat((size_t)(po[1] - po[0]), '\0');
  if (!pat.empty()) {
    hipError_t e = hipMemcpyAsync(&pat[0], (const uint8_t*)pattern->values + po[0], pat.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) {
      ah_out_free(ctx, vals, bytes);
      return ah_fail(ctx, AH_HIP_ERROR, "reading the pattern failed: %s", hipGetErrorString(e));
    }
  }
  std::vector<uint16_t> prog;
  if (op == AH_LIKE || op == AH_NLIKE) {
    tokenize_like(pat, &prog);
  } else {  // starts_with / ends_with / contains take the needle literally (like.rs:138-205)
    if (op == AH_ENDS_WITH || op == AH_CONTAINS) prog.push_back(TOK_STAR);
    for (unsigned char c : pat) prog.push_back(c);
    if (op == AH_STARTS_WITH || op == AH_CONTAINS) prog.push_back(TOK_STAR);
  }
  uint16_t* dprog = nullptr;
  ah_status st = ah_pool_alloc(ctx, std::max<size_t>(prog.size(), 1) * 2, (void**)&dprog);
  if (st == AH_OK && values->validity) st = ah_out_alloc(ctx, bytes, (void**)&nb);
  if (st != AH_OK) {
    ah_pool_free(ctx, dprog);
    ah_out_free(ctx, vals, bytes);
    return st;
  }
  if (!prog.empty()) hipMemcpyAsync(dprog, prog.data(), prog.size() * 2, hipMemcpyHostToDevice, ctx->stream);
  {
    ah_prof_scope ps(ctx, "string_like");
    const int64_t nwords = (len + 63) >> 6;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(256 * 32, ah_ceil_div(nwords, 4)));
    const int neg = op == AH_NLIKE;
    if (t == AH_UTF8)
      like_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t*)values->offsets, (const uint8_t*)values->values, len,
                                                          dprog, (int)prog.size(), neg, vals);
    else
      like_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t*)values->offsets, (const uint8_t*)values->values, len,
                                                          dprog, (int)prog.size(), neg, vals);
  }
  int64_t set = len;
  if (nb)  // from_unary: the input's nulls are cloned (presence-based)
    st = ah_bitmap_op(ctx, BM_COPY, make_bitview(values->validity, values->validity_bit_offset), BitView{nullptr, 0},
                      BitView{nullptr, 0}, len, nb, &set);
  hipError_t e = hipGetLastError();
  if (st == AH_OK && e == hipSuccess) e = ah_stream_wait(ctx);  // prog (host vector) must outlive the H2D copy
  ah_pool_free(ctx, dprog);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, vals, bytes);
    ah_out_free(ctx, nb, bytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "string predicate failed: %s", hipGetErrorString(e));
  }
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (nb) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len - set;
  }
  return AH_OK;
}

// arrow_string::length::length (arrow-string/src/length.rs:58-110) for Utf8 / LargeUtf8: the byte length of every
// value as Int32 / Int64 (the offset type), nulls cloned; bit_length = 8 x that (:130).
namespace {
template <typename O>
__global__ __launch_bounds__(256) void length_kernel(const O* offs, int64_t len, int mul, O* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256)
    out[i] = (O)((offs[i + 1] - offs[i]) * (O)mul);
}
}  // namespace

extern "C" ah_status ah_string_length(ah_context* ctx, const ah_array_view* values, int32_t bits, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  const ah_type t = values->type;
  if (!(t == AH_UTF8 || t == AH_LARGE_UTF8))
    return ah_fail(ctx, AH_COMPUTE_ERROR, "length not supported for %s", ah_type_name(t));  // length.rs:104-107
  if (!values->offsets && values->length) return ah_fail(ctx, AH_INVALID_ARGUMENT, "string array view without offsets");
  const int64_t len = values->length;
  out->type = t == AH_UTF8 ? AH_INT32 : AH_INT64;
  out->length = len;
  if (len == 0) return AH_OK;
  const size_t ow = t == AH_UTF8 ? 4 : 8, vbytes = (size_t)len * ow, bbytes = ah_bitmap_bytes(len);
  void* ov = nullptr;
  unsigned long long* nb = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  if (values->validity) {
    ah_status st = ah_out_alloc(ctx, bbytes, (void**)&nb);
    if (st != AH_OK) {
      ah_out_free(ctx, ov, vbytes);
      return st;
    }
  }
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(len, 256), 8192));
  if (t == AH_UTF8) length_kernel<int32_t><<<grid, 256, 0, ctx->stream>>>((const int32_t*)values->offsets, len, bits ? 8 : 1, (int32_t*)ov);
  else length_kernel<int64_t><<<grid, 256, 0, ctx->stream>>>((const int64_t*)values->offsets, len, bits ? 8 : 1, (int64_t*)ov);
  int64_t set = len;
  ah_status st = AH_OK;
  if (nb)
    st = ah_bitmap_op(ctx, BM_COPY, make_bitview(values->validity, values->validity_bit_offset), BitView{nullptr, 0},
                      BitView{nullptr, 0}, len, nb, &set);
  hipError_t e = hipGetLastError();
  if (st == AH_OK && e == hipSuccess) e = ah_stream_wait(ctx);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, nb, bbytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "length kernel failed: %s", hipGetErrorString(e));
  }
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (nb) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = len - set;
  }
  return AH_OK;
}
