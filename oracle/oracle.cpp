// oracle.cpp — CPU restatement of the arrow-rs compute hot path (see oracle.h).
// TEST INFRASTRUCTURE ONLY: parity checker + bench.py cpu_baseline ("port").
// Every function cites the reference file:line (under /root/reference) whose
// algorithm and edge semantics it follows.  Scalar, single-threaded, built
// -O3 -march=native so the compiler auto-vectorises the same loops LLVM does
// for the reference (arrow/README.md:147-170).
#include "oracle.h"

#include <algorithm>
#include <functional>
#include <charconv>
#include <cinttypes>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

int32_t fail(int32_t st, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return st;
}

int type_width(int32_t t) {
  switch (t) {
    case ORC_BOOL: return 0;
    case ORC_INT8: case ORC_UINT8: return 1;
    case ORC_INT16: case ORC_UINT16: case ORC_FLOAT16: return 2;
    case ORC_INT32: case ORC_UINT32: case ORC_FLOAT32: return 4;
    case ORC_INT64: case ORC_UINT64: case ORC_FLOAT64: return 8;
    case ORC_FIXED16: return 16;
    case ORC_FIXED32: return 32;
    default: return -1;
  }
}
const char* type_name(int32_t t) {
  switch (t) {
    case ORC_BOOL: return "Boolean";
    case ORC_INT8: return "Int8"; case ORC_INT16: return "Int16";
    case ORC_INT32: return "Int32"; case ORC_INT64: return "Int64";
    case ORC_UINT8: return "UInt8"; case ORC_UINT16: return "UInt16";
    case ORC_UINT32: return "UInt32"; case ORC_UINT64: return "UInt64";
    case ORC_FLOAT16: return "Float16"; case ORC_FLOAT32: return "Float32";
    case ORC_FLOAT64: return "Float64";
    case ORC_FIXED16: return "FixedWidth16"; case ORC_FIXED32: return "FixedWidth32";
    case ORC_UTF8: return "Utf8"; case ORC_LARGE_UTF8: return "LargeUtf8";
    default: return "?";
  }
}
bool is_integer(int32_t t) { return t >= ORC_INT8 && t <= ORC_UINT64; }
bool is_signed_int(int32_t t) { return t >= ORC_INT8 && t <= ORC_INT64; }

// ----------------------------------------------------------------- bit_util
// arrow-buffer/src/util/bit_util.rs:63 get_bit_raw / set_bit_raw
inline bool get_bit(const uint8_t* d, int64_t i) { return (d[i >> 3] >> (i & 7)) & 1; }
inline void set_bit(uint8_t* d, int64_t i) { d[i >> 3] |= (uint8_t)(1u << (i & 7)); }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t bitmap_bytes(int64_t bits) { return (size_t)ceil_div(bits, 64) * 8; }

// UnalignedBitChunk::new (arrow-buffer/src/util/bit_chunk_iterator.rs:41-132):
// optional prefix word, 8-byte-aligned middle words, optional suffix word, with
// the lead/trailing padding bits masked off.
struct UnalignedBitChunk {
  int64_t lead_padding = 0, trailing_padding = 0;
  bool has_prefix = false, has_suffix = false;
  uint64_t prefix = 0, suffix = 0;
  const uint64_t* chunks = nullptr;
  int64_t nchunks = 0;

  static uint64_t read_u64(const uint8_t* p, size_t n) {  // :176-181
    uint64_t v = 0;
    memcpy(&v, p, n < 8 ? n : 8);
    return v;
  }
  static uint64_t prefix_mask(int64_t lead) { return ~((1ull << lead) - 1); }  // :184-186
  static void suffix_mask(int64_t len, int64_t lead, uint64_t* mask, int64_t* trailing) {  // :189-199
    int64_t trailing_bits = (len + lead) % 64;
    if (trailing_bits == 0) {
      *mask = ~0ull;
      *trailing = 0;
      return;
    }
    *trailing = 64 - trailing_bits;
    *mask = (1ull << trailing_bits) - 1;
  }

  UnalignedBitChunk(const uint8_t* buffer, int64_t offset, int64_t len) {
    if (len == 0) return;
    int64_t byte_offset = offset / 8, offset_padding = offset % 8;
    int64_t bytes_len = ceil_div(len + offset_padding, 8);
    const uint8_t* buf = buffer + byte_offset;
    uint64_t pmask = prefix_mask(offset_padding);
    if (bytes_len <= 8) {
      uint64_t smask;
      suffix_mask(len, offset_padding, &smask, &trailing_padding);
      prefix = read_u64(buf, (size_t)bytes_len) & smask & pmask;
      has_prefix = true;
      lead_padding = offset_padding;
      return;
    }
    if (bytes_len <= 16) {
      uint64_t smask;
      suffix_mask(len, offset_padding, &smask, &trailing_padding);
      prefix = read_u64(buf, 8) & pmask;
      suffix = read_u64(buf + 8, (size_t)(bytes_len - 8)) & smask;
      has_prefix = has_suffix = true;
      lead_padding = offset_padding;
      return;
    }
    // align_to::<u64>()
    uintptr_t addr = (uintptr_t)buf;
    size_t pre = (size_t)((8 - (addr & 7)) & 7);
    size_t mid = ((size_t)bytes_len - pre) / 8;
    size_t suf = (size_t)bytes_len - pre - mid * 8;
    chunks = (const uint64_t*)(buf + pre);
    nchunks = (int64_t)mid;
    int64_t alignment_padding = 0;
    if (offset_padding == 0 && pre == 0) {
      // no prefix
    } else if (pre == 0) {
      prefix = chunks[0] & pmask;
      has_prefix = true;
      chunks += 1;
      nchunks -= 1;
    } else {
      alignment_padding = (int64_t)(8 - pre) * 8;
      prefix = (read_u64(buf, pre) & pmask) << alignment_padding;
      has_prefix = true;
    }
    lead_padding = offset_padding + alignment_padding;
    uint64_t smask;
    suffix_mask(len, lead_padding, &smask, &trailing_padding);
    if (trailing_padding == 0) {
      // the suffix bytes (if any) cannot exist when trailing_padding == 0
    } else if (suf == 0) {
      suffix = chunks[nchunks - 1] & smask;
      has_suffix = true;
      nchunks -= 1;
    } else {
      suffix = read_u64(buf + pre + mid * 8, suf) & smask;
      has_suffix = true;
    }
  }
  int64_t nwords() const { return (has_prefix ? 1 : 0) + nchunks + (has_suffix ? 1 : 0); }
  uint64_t word(int64_t i) const {
    if (has_prefix) {
      if (i == 0) return prefix;
      i -= 1;
    }
    if (i < nchunks) return chunks[i];
    return suffix;
  }
  int64_t count_ones() const {  // :169-171
    int64_t c = 0;
    if (has_prefix) c += __builtin_popcountll(prefix);
    for (int64_t i = 0; i < nchunks; ++i) c += __builtin_popcountll(chunks[i]);
    if (has_suffix) c += __builtin_popcountll(suffix);
    return c;
  }
};

// BitIndexIterator (arrow-buffer/src/util/bit_iterator.rs:284-324)
struct BitIndexIterator {
  UnalignedBitChunk c;
  int64_t wi = 0, nw;
  uint64_t current_chunk;
  int64_t chunk_offset;
  BitIndexIterator(const uint8_t* b, int64_t off, int64_t len) : c(b, off, len) {
    nw = c.nwords();
    current_chunk = nw > 0 ? c.word(0) : 0;
    wi = 1;
    chunk_offset = -c.lead_padding;
  }
  inline bool next(int64_t* out) {
    for (;;) {
      if (current_chunk != 0) {
        int bit_pos = __builtin_ctzll(current_chunk);
        current_chunk &= current_chunk - 1;
        *out = chunk_offset + bit_pos;
        return true;
      }
      if (wi >= nw) return false;
      current_chunk = c.word(wi++);
      chunk_offset += 64;
    }
  }
};

// BitSliceIterator (arrow-buffer/src/util/bit_iterator.rs:188-277)
struct BitSliceIterator {
  UnalignedBitChunk c;
  int64_t wi = 0, nw, len;
  int64_t current_offset;
  uint64_t current_chunk;
  BitSliceIterator(const uint8_t* b, int64_t off, int64_t l) : c(b, off, l), len(l) {
    nw = c.nwords();
    current_offset = -c.lead_padding;
    current_chunk = nw > 0 ? c.word(0) : 0;
    wi = 1;
  }
  bool advance_to_set_bit(int64_t* chunk_off, int* bit) {
    for (;;) {
      if (current_chunk != 0) {
        *bit = __builtin_ctzll(current_chunk);
        *chunk_off = current_offset;
        return true;
      }
      if (wi >= nw) return false;
      current_chunk = c.word(wi++);
      current_offset += 64;
    }
  }
  bool next(int64_t* start, int64_t* end) {
    if (len == 0) return false;
    int64_t start_chunk;
    int start_bit;
    if (!advance_to_set_bit(&start_chunk, &start_bit)) return false;
    current_chunk |= (1ull << start_bit) - 1;
    for (;;) {
      if (current_chunk != ~0ull) {
        int end_bit = __builtin_ctzll(~current_chunk);  // trailing_ones
        current_chunk &= ~((1ull << end_bit) - 1);
        *start = start_chunk + start_bit;
        *end = current_offset + end_bit;
        return true;
      }
      if (wi < nw) {
        current_chunk = c.word(wi++);
        current_offset += 64;
      } else {
        *start = start_chunk + start_bit;
        *end = len;
        len = 0;
        return true;
      }
    }
  }
};

int64_t count_set_bits(const uint8_t* bits, int64_t off, int64_t len) {
  if (!bits) return len;
  return UnalignedBitChunk(bits, off, len).count_ones();
}

// BooleanBufferBuilder::append_packed_range (arrow-buffer/src/builder/boolean.rs:287-301)
// via bit_mask::set_bits (arrow-buffer/src/util/bit_mask.rs:33): copy `len` bits.
void copy_bits(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t len) {
  for (int64_t i = 0; i < len; ++i)
    if (get_bit(src, src_off + i)) set_bit(dst, dst_off + i);
}

void* xalloc(size_t n) {
  void* p = calloc(n ? n : 8, 1);
  if (!p) abort();
  return p;
}

void out_init(orc_out* o) { memset(o, 0, sizeof *o); }

int64_t resolve_nulls(const orc_view* v) {
  if (!v->validity) return 0;
  if (v->null_count >= 0) return v->null_count;
  return v->length - count_set_bits(v->validity, v->validity_bit_offset, v->length);
}

// ------------------------------------------------------------------- filter
enum Strategy { S_NONE, S_ALL, S_SLICES, S_INDICES };

struct Predicate {  // FilterPredicate (arrow-select/src/filter.rs:442-449)
  std::vector<uint8_t> owned;
  const uint8_t* bits = nullptr;
  int64_t off = 0, len = 0, count = 0;
  Strategy strategy = S_NONE;
};

// `avail` (1..64) bits of a bitmap starting at bit `off`, in the low bits of the result (the rest 0); reads only bytes that hold them
static inline uint64_t bits64(const uint8_t* b, int64_t off, int avail) {
  const uint8_t* p = b + (off >> 3);
  const int sh = (int)(off & 7), nbytes = (sh + avail + 7) >> 3;  // <= 9
  uint64_t lo = 0;
  memcpy(&lo, p, (size_t)(nbytes < 8 ? nbytes : 8));
  uint64_t x = lo >> sh;
  if (nbytes == 9) x |= (uint64_t)p[8] << (64 - sh);
  return avail == 64 ? x : x & ((1ull << avail) - 1);
}

// BooleanArray::true_count (arrow-array/src/array/boolean_array.rs:175-187): with a null buffer the reference ANDs values and
// validity a word at a time (buffer_bin_and + count_set_bits), so does this — the per-bit loop it replaces cost the CPU
// baseline of the "filter context w NULLs" shapes a multiple of the reference's own time
int64_t true_count(const orc_view* p) {
  if (p->length == 0) return 0;
  if (!p->validity) return count_set_bits((const uint8_t*)p->values, p->values_bit_offset, p->length);
  int64_t c = 0;
  const uint8_t* v = (const uint8_t*)p->values;
  for (int64_t i = 0; i < p->length; i += 64) {
    const int n = (int)std::min<int64_t>(64, p->length - i);
    c += __builtin_popcountll(bits64(v, p->values_bit_offset + i, n) & bits64(p->validity, p->validity_bit_offset + i, n));
  }
  return c;
}

// FilterBuilder::new_with_count (filter.rs:260-273) + default_strategy (:346-364)
void build_predicate(const orc_view* p, Predicate* out) {
  out->len = p->length;
  out->count = true_count(p);
  if (p->validity && resolve_nulls(p) != 0) {
    // prep_null_mask_filter (filter.rs:167-171): values & validity, offset 0
    out->owned.assign(bitmap_bytes(p->length), 0);
    const uint8_t* v = (const uint8_t*)p->values;
    for (int64_t i = 0; i < p->length; i += 64) {  // buffer_bin_and (filter.rs:169): a word at a time
      const int n = (int)std::min<int64_t>(64, p->length - i);
      const uint64_t w = bits64(v, p->values_bit_offset + i, n) & bits64(p->validity, p->validity_bit_offset + i, n);
      memcpy(out->owned.data() + (i >> 3), &w, (size_t)((n + 7) >> 3));
    }
    out->bits = out->owned.data();
    out->off = 0;
  } else {
    out->bits = (const uint8_t*)p->values;
    out->off = p->values_bit_offset;
  }
  if (out->len == 0 || out->count == 0) out->strategy = S_NONE;
  else if (out->count == out->len) out->strategy = S_ALL;
  else if ((double)out->count / (double)out->len > 0.8) out->strategy = S_SLICES;  // :43
  else out->strategy = S_INDICES;
}

// filter_native (filter.rs:731-770)
template <typename T>
void filter_native_t(const T* values, const Predicate& p, T* out) {
  if (p.strategy == S_SLICES) {
    BitSliceIterator it(p.bits, p.off, p.len);
    int64_t s, e;
    T* o = out;
    while (it.next(&s, &e)) {
      memcpy(o, values + s, (size_t)(e - s) * sizeof(T));
      o += e - s;
    }
  } else {
    BitIndexIterator it(p.bits, p.off, p.len);
    int64_t idx = 0;
    for (int64_t i = 0; i < p.count; ++i) {  // IndexIterator is trusted-len (:84-138)
      it.next(&idx);
      out[i] = values[idx];
    }
  }
}
struct B16 { uint64_t a, b; };
struct B32 { uint64_t a, b, c, d; };

void filter_native(const void* values, int width, const Predicate& p, void* out) {
  switch (width) {
    case 1: filter_native_t((const uint8_t*)values, p, (uint8_t*)out); break;
    case 2: filter_native_t((const uint16_t*)values, p, (uint16_t*)out); break;
    case 4: filter_native_t((const uint32_t*)values, p, (uint32_t*)out); break;
    case 8: filter_native_t((const uint64_t*)values, p, (uint64_t*)out); break;
    case 16: filter_native_t((const B16*)values, p, (B16*)out); break;
    case 32: filter_native_t((const B32*)values, p, (B32*)out); break;
  }
}

// filter_bits (filter.rs:680-720): K bits out, offset 0
void filter_bits(const uint8_t* src, int64_t src_off, const Predicate& p, uint8_t* out) {
  if (p.strategy == S_SLICES) {
    BitSliceIterator it(p.bits, p.off, p.len);
    int64_t s, e, o = 0;
    while (it.next(&s, &e)) {
      copy_bits(out, o, src, src_off + s, e - s);
      o += e - s;
    }
  } else {
    BitIndexIterator it(p.bits, p.off, p.len);
    int64_t idx = 0;
    uint64_t* words = (uint64_t*)out;
    uint64_t packed = 0;
    for (int64_t i = 0; i < p.count; ++i) {  // from_trusted_len_iter_bool
      it.next(&idx);
      packed |= (uint64_t)get_bit(src, src_off + idx) << (i & 63);
      if ((i & 63) == 63) {
        words[i >> 6] = packed;
        packed = 0;
      }
    }
    if (p.count & 63) words[p.count >> 6] = packed;
  }
}

// filter_bytes (filter.rs:890-928): offsets from the selected rows' lengths (extend_offsets_idx /
// extend_offsets_slices), then the bytes (extend_idx / extend_slices); null slots are copied too
template <typename OFF>
void filter_bytes(const orc_view* values, const Predicate& p, orc_out* out) {
  const OFF* so = (const OFF*)values->offsets;
  const uint8_t* sv = (const uint8_t*)values->values;
  OFF* d_off = (OFF*)xalloc((size_t)(p.count + 1) * sizeof(OFF));
  OFF cur = 0;
  int64_t j = 0;
  d_off[0] = 0;
  std::vector<int64_t> rows;
  rows.reserve((size_t)p.count);
  if (p.strategy == S_SLICES) {
    BitSliceIterator it(p.bits, p.off, p.len);
    int64_t s, e;
    while (it.next(&s, &e)) for (int64_t i = s; i < e; ++i) rows.push_back(i);
  } else {
    BitIndexIterator it(p.bits, p.off, p.len);
    int64_t idx = 0;
    for (int64_t i = 0; i < p.count; ++i) {
      it.next(&idx);
      rows.push_back(idx);
    }
  }
  for (int64_t r : rows) {
    cur += so[r + 1] - so[r];
    d_off[++j] = cur;
  }
  uint8_t* dv = (uint8_t*)xalloc((size_t)cur);
  size_t w = 0;
  for (int64_t r : rows) {
    size_t n = (size_t)(so[r + 1] - so[r]);
    memcpy(dv + w, sv + so[r], n);
    w += n;
  }
  out->offsets = d_off;
  out->offsets_bytes = (p.count + 1) * (int64_t)sizeof(OFF);
  out->values = dv;
  out->values_bytes = (int64_t)cur;
}

// FilterPredicate::filter_nulls (filter.rs:512-532)
void filter_nulls(const orc_view* values, const Predicate& p, orc_out* out) {
  if (!values->validity) return;
  if (resolve_nulls(values) == 0) return;
  size_t bytes = bitmap_bytes(p.count);
  uint8_t* nb = (uint8_t*)xalloc(bytes);
  filter_bits(values->validity, values->validity_bit_offset, p, nb);
  int64_t null_count = p.count - count_set_bits(nb, 0, p.count);
  if (null_count == 0) {
    free(nb);
    return;
  }
  out->validity = nb;
  out->validity_bytes = (int64_t)bytes;
  out->null_count = null_count;
}

int32_t filter_impl(const orc_view* values, const orc_view* pred, orc_out* out) {
  out_init(out);
  if (pred->type != ORC_BOOL)
    return fail(ORC_INVALID_ARGUMENT, "filter predicate must be Boolean, got %s", type_name(pred->type));
  // filter_array (filter.rs:535-541) — checked before anything else observable
  if (pred->length > values->length)
    return fail(ORC_INVALID_ARGUMENT,
                "Filter predicate of length %lld is larger than target array of length %lld",
                (long long)pred->length, (long long)values->length);
  const bool is_string = values->type == ORC_UTF8 || values->type == ORC_LARGE_UTF8;
  int width = is_string ? 0 : type_width(values->type);
  if (width < 0) return fail(ORC_NOT_YET_IMPLEMENTED, "filter not supported for type %s", type_name(values->type));
  Predicate p;
  build_predicate(pred, &p);
  out->type = values->type;
  if (p.strategy == S_NONE) {  // :545 new_empty_array
    out->length = 0;
    if (is_string) {
      out->offsets_bytes = values->type == ORC_UTF8 ? 4 : 8;
      out->offsets = xalloc((size_t)out->offsets_bytes);
    }
    return ORC_OK;
  }
  if (p.strategy == S_ALL) {  // :546 values.slice(0, count)
    out->length = p.count;
    out->values = const_cast<void*>(values->values);
    out->values_bit_offset = values->values_bit_offset;
    out->values_bytes = width ? p.count * width : 0;
    out->offsets = const_cast<void*>(values->offsets);
    out->flags = 1;
    if (values->validity) {
      out->validity = const_cast<uint8_t*>(values->validity);
      out->validity_bit_offset = values->validity_bit_offset;
      out->null_count = p.count - count_set_bits(values->validity, values->validity_bit_offset, p.count);
    }
    return ORC_OK;
  }
  out->length = p.count;
  if (is_string) {
    if (values->type == ORC_UTF8) filter_bytes<int32_t>(values, p, out);
    else filter_bytes<int64_t>(values, p, out);
  } else if (values->type == ORC_BOOL) {  // filter_boolean (:723-729)
    size_t bytes = bitmap_bytes(p.count);
    out->values = xalloc(bytes);
    out->values_bytes = (int64_t)bytes;
    filter_bits((const uint8_t*)values->values, values->values_bit_offset, p, (uint8_t*)out->values);
  } else {  // filter_primitive (:773-788)
    out->values = xalloc((size_t)p.count * width);
    out->values_bytes = p.count * width;
    filter_native(values->values, width, p, out->values);
  }
  filter_nulls(values, p, out);
  return ORC_OK;
}

// --------------------------------------------------------------------- take
template <typename I> inline uint64_t to_index(I v) {  // ToIndices (take.rs:1030-1084)
  if constexpr (sizeof(I) <= 2) {
    if constexpr (std::is_signed<I>::value) return (uint32_t)(int32_t)v;  // `as u32` sign-extends
    else return (uint32_t)v;
  } else if constexpr (sizeof(I) == 4) return (uint32_t)v;  // reinterpret
  else return (uint64_t)v;
}

struct TakePanic {
  bool hit = false;
  std::string msg;
};

// take_native (take.rs:432-457)
template <typename T, typename I>
void take_native(const T* values, int64_t values_len, const I* idx, int64_t n, const uint8_t* ivalid,
                 int64_t ivalid_off, int64_t idx_nulls, T* out, TakePanic* panic) {
  if (ivalid && idx_nulls > 0) {
    for (int64_t i = 0; i < n; ++i) {
      uint64_t ix = to_index(idx[i]);
      if (ix < (uint64_t)values_len) out[i] = values[ix];
      else if (!get_bit(ivalid, ivalid_off + i)) out[i] = T{};
      else {
        panic->hit = true;
        char b[64];
        snprintf(b, sizeof b, "Out-of-bounds index %llu", (unsigned long long)ix);
        panic->msg = b;
        return;
      }
    }
  } else {
    for (int64_t i = 0; i < n; ++i) {
      uint64_t ix = to_index(idx[i]);
      if (ix >= (uint64_t)values_len) {
        panic->hit = true;
        char b[96];
        snprintf(b, sizeof b, "index out of bounds: the len is %lld but the index is %llu",
                 (long long)values_len, (unsigned long long)ix);
        panic->msg = b;
        return;
      }
      out[i] = values[ix];
    }
  }
}

// take_bits (take.rs:459-486); `values.value(i)` asserts i < bit_len (boolean.rs:495)
template <typename I>
void take_bits(const uint8_t* bits, int64_t bits_off, int64_t bits_len, const I* idx, int64_t n,
               const uint8_t* ivalid, int64_t ivalid_off, int64_t idx_nulls, uint8_t* out,
               TakePanic* panic) {
  if (ivalid && idx_nulls > 0) {
    for (int64_t i = 0; i < n; ++i) {  // nulls.valid_indices()
      if (!get_bit(ivalid, ivalid_off + i)) continue;
      uint64_t ix = to_index(idx[i]);
      if (ix >= (uint64_t)bits_len) {
        panic->hit = true;
        panic->msg = "assertion failed: idx < self.bit_len";
        return;
      }
      if (get_bit(bits, bits_off + (int64_t)ix)) set_bit(out, i);
    }
  } else {
    for (int64_t i = 0; i < n; ++i) {  // collect_bool
      uint64_t ix = to_index(idx[i]);
      if (ix >= (uint64_t)bits_len) {
        panic->hit = true;
        panic->msg = "assertion failed: idx < self.bit_len";
        return;
      }
      if (get_bit(bits, bits_off + (int64_t)ix)) set_bit(out, i);
    }
  }
}

// check_bounds (take.rs:167-209)
template <typename I>
int32_t check_bounds(int64_t len, const I* idx, int64_t n, const uint8_t* ivalid, int64_t ivalid_off,
                     int64_t idx_nulls) {
  // T::Native::from_usize(len) :174
  if ((uint64_t)len > (uint64_t)std::numeric_limits<I>::max()) return ORC_OK;
  I l = (I)len;
  for (int64_t i = 0; i < n; ++i) {
    if (idx_nulls > 0 && ivalid && !get_bit(ivalid, ivalid_off + i)) continue;
    I v = idx[i];
    bool bad = v >= l;
    if constexpr (std::is_signed<I>::value) bad = bad || (idx_nulls == 0 && v < 0);
    // note: with index nulls the reference only tests `index >= len` (:184)
    if (bad) {
      if constexpr (std::is_signed<I>::value)
        return fail(ORC_COMPUTE_ERROR,
                    "Array index out of bounds, cannot get item at index %lld from %lld entries",
                    (long long)v, (long long)len);
      else
        return fail(ORC_COMPUTE_ERROR,
                    "Array index out of bounds, cannot get item at index %llu from %lld entries",
                    (unsigned long long)v, (long long)len);
    }
  }
  return ORC_OK;
}

// take_bytes (take.rs:499-627)
template <typename OFF, typename I>
int32_t take_bytes(const orc_view* values, const orc_view* indices, orc_out* out) {
  const I* idx = (const I*)indices->values;
  const int64_t n = indices->length;
  const int64_t idx_nulls = resolve_nulls(indices);
  const uint8_t* iv = indices->validity;
  const int64_t ivo = indices->validity_bit_offset;
  const OFF* so = (const OFF*)values->offsets;
  const uint8_t* sv = (const uint8_t*)values->values;
  TakePanic panic;
  // take_nulls (take.rs:418-430)
  uint8_t* ob = nullptr;
  size_t bbytes = bitmap_bytes(n);
  int64_t out_nulls = 0;
  if (values->validity && resolve_nulls(values) > 0) {
    ob = (uint8_t*)xalloc(bbytes);
    take_bits<I>(values->validity, values->validity_bit_offset, values->length, idx, n, iv, ivo, idx_nulls, ob, &panic);
    if (panic.hit) {
      free(ob);
      return fail(ORC_PANIC, "%s", panic.msg.c_str());
    }
    out_nulls = n - count_set_bits(ob, 0, n);
    if (out_nulls == 0) {
      free(ob);
      ob = nullptr;
    }
  } else if (iv) {
    ob = (uint8_t*)xalloc(bbytes);
    copy_bits(ob, 0, iv, ivo, n);
    out_nulls = idx_nulls;
  }
  OFF* d_off = (OFF*)xalloc((size_t)(n + 1) * sizeof(OFF));
  uint64_t capacity = 0;
  std::vector<std::pair<int64_t, int64_t>> ranges;
  const int64_t off_len = values->length + 1;  // input_offsets.len()
  for (int64_t i = 0; i < n; ++i) {
    bool live = out_nulls == 0 || get_bit(ob, i);  // nullable path only visits valid output slots
    if (live) {
      uint64_t ix = to_index(idx[i]);
      // input_offsets[index] then input_offsets[index + 1] (bounds-checked slice indexing)
      uint64_t bad = ix >= (uint64_t)off_len ? ix : (ix + 1 >= (uint64_t)off_len ? ix + 1 : ~0ull);
      if (bad != ~0ull) {
        free(ob);
        free(d_off);
        return fail(ORC_PANIC, "index out of bounds: the len is %lld but the index is %llu", (long long)off_len,
                    (unsigned long long)bad);
      }
      capacity += (uint64_t)(so[ix + 1] - so[ix]);
      if (sizeof(OFF) == 4 && capacity > (uint64_t)INT32_MAX) {
        free(ob);
        free(d_off);
        return fail(ORC_OFFSET_OVERFLOW_ERROR, "%llu", (unsigned long long)capacity);
      }
      ranges.emplace_back((int64_t)so[ix], (int64_t)so[ix + 1]);
    }
    d_off[i + 1] = (OFF)capacity;
  }
  uint8_t* dv = (uint8_t*)xalloc((size_t)capacity);
  size_t w = 0;
  for (auto& r : ranges) {
    memcpy(dv + w, sv + r.first, (size_t)(r.second - r.first));
    w += (size_t)(r.second - r.first);
  }
  out->length = n;
  out->offsets = d_off;
  out->offsets_bytes = (n + 1) * (int64_t)sizeof(OFF);
  out->values = dv;
  out->values_bytes = (int64_t)capacity;
  if (ob) {
    out->validity = ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = out_nulls;
  }
  return ORC_OK;
}

template <typename I>
int32_t take_typed(const orc_view* values, const orc_view* indices, int32_t cb, orc_out* out) {
  const I* idx = (const I*)indices->values;
  int64_t n = indices->length;
  int64_t idx_nulls = resolve_nulls(indices);
  const uint8_t* iv = indices->validity;
  int64_t ivo = indices->validity_bit_offset;
  if (cb) {
    int32_t st = check_bounds<I>(values->length, idx, n, iv, ivo, idx_nulls);
    if (st != ORC_OK) return st;
  }
  out->type = values->type;
  const bool is_string = values->type == ORC_UTF8 || values->type == ORC_LARGE_UTF8;
  if (n == 0) {  // take_impl :215-217
    out->length = 0;
    if (is_string) {
      out->offsets_bytes = values->type == ORC_UTF8 ? 4 : 8;
      out->offsets = xalloc((size_t)out->offsets_bytes);
    }
    return ORC_OK;
  }
  if (is_string)
    return values->type == ORC_UTF8 ? take_bytes<int32_t, I>(values, indices, out)
                                    : take_bytes<int64_t, I>(values, indices, out);
  int width = type_width(values->type);
  TakePanic panic;
  size_t vbytes = width ? (size_t)n * width : bitmap_bytes(n);
  void* ov = xalloc(vbytes);
  switch (width) {
    case 0:  // take_boolean :489-496
      take_bits<I>((const uint8_t*)values->values, values->values_bit_offset, values->length, idx, n,
                   iv, ivo, idx_nulls, (uint8_t*)ov, &panic);
      break;
    case 1: take_native((const uint8_t*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (uint8_t*)ov, &panic); break;
    case 2: take_native((const uint16_t*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (uint16_t*)ov, &panic); break;
    case 4: take_native((const uint32_t*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (uint32_t*)ov, &panic); break;
    case 8: take_native((const uint64_t*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (uint64_t*)ov, &panic); break;
    case 16: take_native((const B16*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (B16*)ov, &panic); break;
    case 32: take_native((const B32*)values->values, values->length, idx, n, iv, ivo, idx_nulls, (B32*)ov, &panic); break;
  }
  if (panic.hit) {
    free(ov);
    return fail(ORC_PANIC, "%s", panic.msg.c_str());
  }
  // take_nulls (take.rs:418-430)
  int64_t val_nulls = resolve_nulls(values);
  uint8_t* ob = nullptr;
  size_t bbytes = bitmap_bytes(n);
  int64_t out_nulls = 0;
  if (values->validity && val_nulls > 0) {
    ob = (uint8_t*)xalloc(bbytes);
    take_bits<I>(values->validity, values->validity_bit_offset, values->length, idx, n, iv, ivo,
                 idx_nulls, ob, &panic);
    if (panic.hit) {
      free(ov);
      free(ob);
      return fail(ORC_PANIC, "%s", panic.msg.c_str());
    }
    out_nulls = n - count_set_bits(ob, 0, n);
    if (out_nulls == 0) {  // from_unsliced_buffer (null.rs:266-270)
      free(ob);
      ob = nullptr;
    }
  } else if (iv) {  // indices.nulls().cloned() (:428) — realigned to offset 0
    ob = (uint8_t*)xalloc(bbytes);
    copy_bits(ob, 0, iv, ivo, n);
    out_nulls = idx_nulls;
  }
  out->length = n;
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (ob) {
    out->validity = ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = out_nulls;
  }
  return ORC_OK;
}

// ------------------------------------------------------------------ Float16
// `half::f16` (crate `half` 2.7.1 per /root/reference/Cargo.lock; a third-party dependency, not vendored in the reference tree —
// the conversion routines below restate its published software fallbacks, which are what arrow-rs gets on x86-64: it depends on
// `half` with default-features = false, arrow-buffer/Cargo.toml:46, so there is no runtime F16C detection).
//   * f16 -> f32 (`f16_to_f32_fallback`): exact; a NaN keeps its sign and payload (shifted left 13) and gets the quiet bit.
//   * f32 -> f16 (`f32_to_f16_fallback`): round to nearest even, overflow -> infinity, results below half the smallest
//     subnormal -> signed zero; a NaN becomes sign | 0x7C00 | 0x0200 | (mantissa >> 13).
//   * arithmetic (`impl Add/Sub/Mul/Div/Rem for f16`): convert both operands to f32, one f32 operation, convert back — never
//     a native half-precision instruction.  Neg flips bit 15.  total_cmp works on the 16-bit pattern like f32's.
//   * num_traits casts (half/src/num_traits.rs): `NumCast for f16` is `n.to_f32().map(f16::from_f32)` (every source type goes
//     through f32: i64 -> f32 `as`, f64 -> f32 `as`, then one more rounding), `ToPrimitive for f16` is `self.to_f32().to_X()`.
// Pinned against numpy's float16 (IEEE binary16, the same round-to-nearest-even conversions) in tests/test_oracle_golden.py.
struct F16 {
  uint16_t bits;
};
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t exp = h & 0x7C00u, man = h & 0x03FFu;
  uint32_t x;
  if (exp == 0x7C00u) x = man ? (sign | 0x7FC00000u | (man << 13)) : (sign | 0x7F800000u);
  else if (exp == 0) {
    if (man == 0) x = sign;
    else {  // subnormal: normalise
      int e = 0;
      uint32_t m = man;
      while (!(m & 0x0400u)) { m <<= 1; ++e; }
      x = sign | ((uint32_t)(127 - 15 - e + 1) << 23) | ((m & 0x03FFu) << 13);
    }
  } else x = sign | (((exp >> 10) + (127 - 15)) << 23) | (man << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}
inline uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = x & 0x80000000u, exp = x & 0x7F800000u, man = x & 0x007FFFFFu;
  if (exp == 0x7F800000u) return (uint16_t)((sign >> 16) | 0x7C00u | (man ? 0x0200u : 0u) | (man >> 13));
  const uint32_t hs = sign >> 16;
  const int he = (int)(exp >> 23) - 127 + 15;
  if (he >= 0x1F) return (uint16_t)(hs | 0x7C00u);
  if (he <= 0) {
    if (14 - he > 24) return (uint16_t)hs;
    const uint32_t m = man | 0x00800000u;
    uint32_t hm = m >> (14 - he);
    const uint32_t round_bit = 1u << (13 - he);
    if ((m & round_bit) != 0 && (m & (3 * round_bit - 1)) != 0) hm += 1;
    return (uint16_t)(hs | hm);
  }
  const uint32_t hm = man >> 13, round_bit = 0x00001000u;
  uint32_t r = hs | ((uint32_t)he << 10) | hm;
  if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) r += 1;
  return (uint16_t)r;
}
template <typename T> struct is_fp : std::is_floating_point<T> {};
template <> struct is_fp<F16> : std::true_type {};

// -------------------------------------------------------------------- arith
enum { OP_ADD = 0, OP_ADD_W, OP_SUB, OP_SUB_W, OP_MUL, OP_MUL_W, OP_DIV, OP_REM };
const char* op_sym(int op) {  // Display for Op (numeric.rs:203-213)
  switch (op) {
    case OP_ADD: case OP_ADD_W: return "+";
    case OP_SUB: case OP_SUB_W: return "-";
    case OP_MUL: case OP_MUL_W: return "*";
    case OP_DIV: return "/";
    default: return "%";
  }
}

template <typename T> std::string dbg(T v) {  // {:?} of a Rust integer
  char b[32];
  if constexpr (std::is_signed<T>::value) snprintf(b, sizeof b, "%lld", (long long)v);
  else snprintf(b, sizeof b, "%llu", (unsigned long long)v);
  return b;
}

// ArrowNativeTypeOp for integers (arrow-array/src/arithmetic.rs:147-284).
// Returns ORC_OK or sets the error exactly like the reference's message.
template <typename T>
inline int32_t int_checked(int op, T l, T r, T* out) {
  bool ovf = false;
  switch (op) {
    case OP_ADD: ovf = __builtin_add_overflow(l, r, out); break;
    case OP_SUB: ovf = __builtin_sub_overflow(l, r, out); break;
    case OP_MUL: ovf = __builtin_mul_overflow(l, r, out); break;
    case OP_DIV:
      if (r == 0) return fail(ORC_DIVIDE_BY_ZERO, "Divide by zero error");
      if (std::is_signed<T>::value && l == std::numeric_limits<T>::min() && r == (T)-1) ovf = true;
      else *out = (T)(l / r);
      break;
    case OP_REM:  // numeric.rs:345-351: zero check, then mod_wrapping
      if (r == 0) return fail(ORC_DIVIDE_BY_ZERO, "Divide by zero error");
      if (std::is_signed<T>::value && r == (T)-1) *out = 0;  // wrapping_rem: MIN % -1 == 0
      else *out = (T)(l % r);
      return ORC_OK;
  }
  if (ovf)
    return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %s %s %s", dbg(l).c_str(), op_sym(op),
                dbg(r).c_str());
  return ORC_OK;
}
template <typename T>
inline T int_wrapping(int op, T l, T r) {
  using U = typename std::make_unsigned<T>::type;
  switch (op) {
    case OP_ADD_W: return (T)((U)l + (U)r);
    case OP_SUB_W: return (T)((U)l - (U)r);
    default: return (T)((U)l * (U)r);  // OP_MUL_W
  }
}
template <typename T>
inline T float_op(int op, T l, T r) {  // arithmetic.rs:308-430, numeric.rs:357-374
  switch (op) {
    case OP_ADD: case OP_ADD_W: return l + r;
    case OP_SUB: case OP_SUB_W: return l - r;
    case OP_MUL: case OP_MUL_W: return l * r;
    case OP_DIV: return l / r;
    default: return std::fmod(l, r);  // Rust `%` on floats == fmod
  }
}

template <>
inline F16 float_op<F16>(int op, F16 l, F16 r) {  // to f32, one f32 operation, one rounding back (half 2.7.1 `impl Add for f16` ...)
  return F16{f32_to_f16(float_op<float>(op, f16_to_f32(l.bits), f16_to_f32(r.bits)))};
}

bool op_is_checked_int(int op) { return op == OP_ADD || op == OP_SUB || op == OP_MUL || op == OP_DIV || op == OP_REM; }

// NullBuffer::union (arrow-buffer/src/buffer/null.rs:79-88): presence-based
uint8_t* nulls_union(const orc_view* a, const orc_view* b, int64_t len, int64_t* null_count) {
  if (!a->validity && !b->validity) return nullptr;
  uint8_t* o = (uint8_t*)xalloc(bitmap_bytes(len));
  for (int64_t i = 0; i < len; ++i) {
    bool va = !a->validity || get_bit(a->validity, a->validity_bit_offset + i);
    bool vb = !b->validity || get_bit(b->validity, b->validity_bit_offset + i);
    if (va && vb) set_bit(o, i);
  }
  *null_count = len - count_set_bits(o, 0, len);
  return o;
}
uint8_t* nulls_clone(const orc_view* a, int64_t len) {
  if (!a->validity) return nullptr;
  uint8_t* o = (uint8_t*)xalloc(bitmap_bytes(len));
  copy_bits(o, 0, a->validity, a->validity_bit_offset, len);
  return o;
}

template <typename T>
int32_t arith_typed(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, orc_out* out) {
  constexpr bool is_float = is_fp<T>::value;
  const bool checked = !is_float && op_is_checked_int(op);
  const T* lv = (const T*)l->values;
  const T* rv = (const T*)r->values;
  out->type = l->type;
  if (l_s == r_s) {
    // arity::binary (arity.rs:104-135) / try_binary (:254-299)
    if (l->length != r->length)
      return fail(ORC_COMPUTE_ERROR, checked ? "Cannot perform a binary operation on arrays of different length"
                                             : "Cannot perform binary operation on arrays of different length");
    int64_t len = l->length;
    out->length = len;
    if (len == 0) return ORC_OK;
    T* ov = (T*)xalloc((size_t)len * sizeof(T));
    out->values = ov;
    out->values_bytes = len * (int64_t)sizeof(T);
    if (!checked) {
      int64_t nc = 0;
      uint8_t* nb = nulls_union(l, r, len, &nc);
      for (int64_t i = 0; i < len; ++i) {
        if constexpr (is_float) ov[i] = float_op<T>(op, lv[i], rv[i]);
        else ov[i] = int_wrapping<T>(op, lv[i], rv[i]);
      }
      if (nb) {
        out->validity = nb;
        out->validity_bytes = (int64_t)bitmap_bytes(len);
        out->null_count = nc;
      }
      return ORC_OK;
    }
    if constexpr (!is_float) {
      bool nullable = resolve_nulls(l) != 0 || resolve_nulls(r) != 0;  // is_nullable()
      int64_t nc = 0;
      uint8_t* nb = nullable ? nulls_union(l, r, len, &nc) : nullptr;
      for (int64_t i = 0; i < len; ++i) {
        if (nb && !get_bit(nb, i)) continue;  // try_for_each_valid_idx: zero elsewhere
        int32_t st = int_checked<T>(op, lv[i], rv[i], &ov[i]);
        if (st != ORC_OK) {
          free(nb);
          orc_release(out);
          return st;
        }
      }
      if (nb) {
        out->validity = nb;
        out->validity_bytes = (int64_t)bitmap_bytes(len);
        out->null_count = nc;
      }
    }
    return ORC_OK;
  }
  // one scalar side: op!/try_op! macros (numeric.rs:278-317)
  const orc_view* arr = l_s ? r : l;
  const orc_view* sc = l_s ? l : r;
  int64_t len = arr->length;
  out->length = len;
  T* ov = (T*)xalloc((size_t)len * sizeof(T));
  out->values = ov;
  out->values_bytes = len * (int64_t)sizeof(T);
  if (resolve_nulls(sc) != 0) {  // PrimitiveArray::new_null(len) (primitive_array.rs:658-664)
    out->validity = (uint8_t*)xalloc(bitmap_bytes(len));
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len;
    return ORC_OK;
  }
  T s = ((const T*)sc->values)[0];
  const T* av = (const T*)arr->values;
  uint8_t* nb = nulls_clone(arr, len);
  for (int64_t i = 0; i < len; ++i) {
    T a = l_s ? s : av[i];
    T b = l_s ? av[i] : s;
    if constexpr (is_float) ov[i] = float_op<T>(op, a, b);  // unary (primitive_array.rs:916-925)
    else if (!checked) ov[i] = int_wrapping<T>(op, a, b);
    else {  // try_unary (:990-1016): valid slots only
      if (nb && !get_bit(nb, i)) continue;
      int32_t st = int_checked<T>(op, a, b, &ov[i]);
      if (st != ORC_OK) {
        free(nb);
        orc_release(out);
        return st;
      }
    }
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  }
  return ORC_OK;
}

// ---------------------------------------------------------------------- cmp
enum { C_EQ = 0, C_NEQ, C_LT, C_LT_EQ, C_GT, C_GT_EQ, C_DISTINCT, C_NOT_DISTINCT };
const char* cmp_sym(int op) {  // Display for Op (arrow-ord/src/cmp.rs:55-68)
  switch (op) {
    case C_EQ: return "==";
    case C_NEQ: return "!=";
    case C_LT: return "<";
    case C_LT_EQ: return "<=";
    case C_GT: return ">";
    case C_GT_EQ: return ">=";
    case C_DISTINCT: return "IS DISTINCT FROM";
    default: return "IS NOT DISTINCT FROM";
  }
}

// ArrowNativeTypeOp::is_lt / is_eq (arrow-array/src/arithmetic.rs:124-127,
// 400-410): floats use IEEE totalOrder and bitwise equality.
template <typename T> inline bool is_lt(T a, T b) { return a < b; }
template <typename T> inline bool is_eq(T a, T b) { return a == b; }
inline int64_t total_key(double x) {
  int64_t b;
  memcpy(&b, &x, 8);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}
inline int32_t total_key(float x) {
  int32_t b;
  memcpy(&b, &x, 4);
  return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
}
inline int16_t total_key(F16 x) {  // f16::total_cmp (half 2.7.1): the f32 rule on the 16-bit pattern
  const int16_t b = (int16_t)x.bits;
  return (int16_t)(b ^ (int16_t)((uint16_t)(b >> 15) >> 1));
}
template <> inline bool is_lt<F16>(F16 a, F16 b) { return total_key(a) < total_key(b); }
template <> inline bool is_eq<F16>(F16 a, F16 b) { return a.bits == b.bits; }
template <> inline bool is_lt<double>(double a, double b) { return total_key(a) < total_key(b); }
template <> inline bool is_lt<float>(float a, float b) { return total_key(a) < total_key(b); }
template <> inline bool is_eq<double>(double a, double b) { return memcmp(&a, &b, 8) == 0; }
template <> inline bool is_eq<float>(float a, float b) { return memcmp(&a, &b, 4) == 0; }

struct BoolVals {  // ArrayOrd for &BooleanArray (cmp.rs:740-761): lt = !l & r
  const uint8_t* bits;
  int64_t off;
};

// collect_bool (cmp.rs:580-611): whole u64 words, optional negation (padding
// bits of the last word are negated too)
template <typename F>
uint8_t* collect_bool(int64_t len, bool neg, F f) {
  uint64_t* buf = (uint64_t*)xalloc(bitmap_bytes(len));
  int64_t chunks = len / 64, rem = len % 64;
  for (int64_t c = 0; c < chunks; ++c) {
    uint64_t packed = 0;
    for (int b = 0; b < 64; ++b) packed |= (uint64_t)f(c * 64 + b) << b;
    buf[c] = neg ? ~packed : packed;
  }
  if (rem) {
    uint64_t packed = 0;
    for (int b = 0; b < rem; ++b) packed |= (uint64_t)f(chunks * 64 + b) << b;
    buf[chunks] = neg ? ~packed : packed;
  }
  return (uint8_t*)buf;
}

// apply_op (cmp.rs:619-648)
template <typename T, typename OP>
uint8_t* apply_op(const T* l, bool l_s, const T* r, bool r_s, int64_t len, bool neg, OP op) {
  if (!l_s && !r_s) return collect_bool(len, neg, [&](int64_t i) { return op(l[i], r[i]); });
  if (l_s && r_s) {
    uint8_t* o = (uint8_t*)xalloc(8);
    if (op(l[0], r[0]) ^ neg) o[0] = 1;
    return o;
  }
  if (l_s) {
    T v = l[0];
    return collect_bool(len, neg, [&](int64_t i) { return op(v, r[i]); });
  }
  T v = r[0];
  return collect_bool(len, neg, [&](int64_t i) { return op(l[i], v); });
}

template <typename T>
uint8_t* cmp_values(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, int64_t len) {
  const T* lv = (const T*)l->values;
  const T* rv = (const T*)r->values;
  auto eq = [](T a, T b) { return is_eq<T>(a, b); };
  auto lt = [](T a, T b) { return is_lt<T>(a, b); };
  switch (op) {  // apply (cmp.rs:480-488)
    case C_EQ: case C_NOT_DISTINCT: return apply_op<T>(lv, l_s, rv, r_s, len, false, eq);
    case C_NEQ: case C_DISTINCT: return apply_op<T>(lv, l_s, rv, r_s, len, true, eq);
    case C_LT: return apply_op<T>(lv, l_s, rv, r_s, len, false, lt);
    case C_LT_EQ: return apply_op<T>(rv, r_s, lv, l_s, len, true, lt);
    case C_GT: return apply_op<T>(rv, r_s, lv, l_s, len, false, lt);
    default: return apply_op<T>(lv, l_s, rv, r_s, len, true, lt);  // C_GT_EQ
  }
}

// ArrayOrd for &[i128] (Decimal128; arrow-ord/src/cmp.rs:713-738): is_eq / is_lt of the native, loaded bytewise
// (host test buffers need not be 16-byte aligned)
uint8_t* cmp_values_i128(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, int64_t len) {
  auto at = [](const orc_view* v, bool sc, int64_t i) { __int128 x; memcpy(&x, (const char*)v->values + (sc ? 0 : i) * 16, 16); return x; };
  auto L = [&](int64_t i) { return at(l, l_s, i); };
  auto R = [&](int64_t i) { return at(r, r_s, i); };
  auto mk = [&](bool neg, auto f) { return collect_bool(len, neg, f); };
  switch (op) {
    case C_EQ: case C_NOT_DISTINCT: return mk(false, [&](int64_t i) { return L(i) == R(i); });
    case C_NEQ: case C_DISTINCT: return mk(true, [&](int64_t i) { return L(i) == R(i); });
    case C_LT: return mk(false, [&](int64_t i) { return L(i) < R(i); });
    case C_LT_EQ: return mk(true, [&](int64_t i) { return R(i) < L(i); });
    case C_GT: return mk(false, [&](int64_t i) { return R(i) < L(i); });
    default: return mk(true, [&](int64_t i) { return L(i) < R(i); });
  }
}

uint8_t* cmp_values_bool(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, int64_t len) {
  const uint8_t* lb = (const uint8_t*)l->values;
  const uint8_t* rb = (const uint8_t*)r->values;
  int64_t lo = l->values_bit_offset, ro = r->values_bit_offset;
  auto L = [&](int64_t i) { return get_bit(lb, lo + (l_s ? 0 : i)); };
  auto R = [&](int64_t i) { return get_bit(rb, ro + (r_s ? 0 : i)); };
  auto mk = [&](bool neg, auto f) { return collect_bool(len, neg, f); };
  switch (op) {
    case C_EQ: case C_NOT_DISTINCT: return mk(false, [&](int64_t i) { return L(i) == R(i); });
    case C_NEQ: case C_DISTINCT: return mk(true, [&](int64_t i) { return L(i) == R(i); });
    case C_LT: return mk(false, [&](int64_t i) { return !L(i) && R(i); });
    case C_LT_EQ: return mk(true, [&](int64_t i) { return !R(i) && L(i); });
    case C_GT: return mk(false, [&](int64_t i) { return !R(i) && L(i); });
    default: return mk(true, [&](int64_t i) { return !L(i) && R(i); });
  }
}

// ArrayOrd for &GenericByteArray (arrow-ord/src/cmp.rs:783-830): is_eq = same bytes, is_lt = bytewise lexicographic
uint8_t* cmp_values_bytes(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, int64_t len) {
  const int ow = l->type == ORC_UTF8 ? 4 : 8;
  auto off = [ow](const orc_view* a, int64_t i) -> int64_t {
    return ow == 4 ? (int64_t)((const int32_t*)a->offsets)[i] : ((const int64_t*)a->offsets)[i];
  };
  auto get = [&](const orc_view* a, bool sc, int64_t i, const uint8_t** p, int64_t* n) {
    const int64_t j = sc ? 0 : i;
    *p = (const uint8_t*)a->values + off(a, j);
    *n = off(a, j + 1) - off(a, j);
  };
  auto eq = [&](const orc_view* a, bool as, const orc_view* b, bool bs, int64_t i) {
    const uint8_t *pa, *pb;
    int64_t na, nb;
    get(a, as, i, &pa, &na);
    get(b, bs, i, &pb, &nb);
    return na == nb && (na == 0 || memcmp(pa, pb, (size_t)na) == 0);
  };
  auto lt = [&](const orc_view* a, bool as, const orc_view* b, bool bs, int64_t i) {
    const uint8_t *pa, *pb;
    int64_t na, nb;
    get(a, as, i, &pa, &na);
    get(b, bs, i, &pb, &nb);
    const int c = std::min(na, nb) ? memcmp(pa, pb, (size_t)std::min(na, nb)) : 0;
    return c < 0 || (c == 0 && na < nb);
  };
  switch (op) {  // apply (cmp.rs:480-488)
    case C_EQ: case C_NOT_DISTINCT: return collect_bool(len, false, [&](int64_t i) { return eq(l, l_s, r, r_s, i); });
    case C_NEQ: case C_DISTINCT: return collect_bool(len, true, [&](int64_t i) { return eq(l, l_s, r, r_s, i); });
    case C_LT: return collect_bool(len, false, [&](int64_t i) { return lt(l, l_s, r, r_s, i); });
    case C_LT_EQ: return collect_bool(len, true, [&](int64_t i) { return lt(r, r_s, l, l_s, i); });
    case C_GT: return collect_bool(len, false, [&](int64_t i) { return lt(r, r_s, l, l_s, i); });
    default: return collect_bool(len, true, [&](int64_t i) { return lt(l, l_s, r, r_s, i); });
  }
}

// ---------------------------------------------------------------------- cast
// num_traits::cast (NumCast/ToPrimitive, num-traits 0.2.19): int->int range
// checked; int->float `as`; float->int succeeds iff trunc(v) fits; float->float `as`.
template <typename I, typename O>
inline bool num_cast(I v, O* out) {
  if constexpr (std::is_floating_point<O>::value) {
    *out = (O)v;
    return true;
  } else if constexpr (std::is_floating_point<I>::value) {
    if (v != v) return false;
    // trunc(v) in [MIN, MAX]
    long double t = std::trunc((long double)v);
    if (t < (long double)std::numeric_limits<O>::min() || t > (long double)std::numeric_limits<O>::max())
      return false;
    *out = (O)v;
    return true;
  } else {
    __int128 x = (__int128)v;
    if (x < (__int128)std::numeric_limits<O>::min() || x > (__int128)std::numeric_limits<O>::max())
      return false;
    *out = (O)v;
    return true;
  }
}

// Float16 on either side (half 2.7.1 num_traits.rs): every conversion goes through f32
template <typename O>
inline bool num_cast_from_f16(F16 v, O* out) {
  if constexpr (std::is_same<O, F16>::value) { *out = v; return true; }
  else return num_cast<float, O>(f16_to_f32(v.bits), out);
}
template <typename I>
inline bool num_cast_to_f16(I v, F16* out) {
  float f;
  num_cast<I, float>(v, &f);  // `as f32`: never fails
  out->bits = f32_to_f16(f);
  return true;
}
template <typename I, typename O>
inline bool num_cast_any(I v, O* out) {
  if constexpr (std::is_same<I, F16>::value) return num_cast_from_f16<O>(v, out);
  else if constexpr (std::is_same<O, F16>::value) return num_cast_to_f16<I>(v, out);
  else return num_cast<I, O>(v, out);
}

template <typename T> std::string dbg_num(T v) {
  if constexpr (std::is_same<T, F16>::value) {  // Debug for f16 prints the f32 value
    char b[40];
    int n = orc_format_f32(f16_to_f32(v.bits), b);
    return std::string(b, (size_t)n);
  } else if constexpr (std::is_floating_point<T>::value) {
    char b[40];
    int n = std::is_same<T, float>::value ? orc_format_f32((float)v, b) : orc_format_f64((double)v, b);
    return std::string(b, (size_t)n);  // Rust {:?} of a float == shortest round-trip, "256.0"
  } else return dbg(v);
}

// cast_numeric_arrays (arrow-cast/src/cast/mod.rs:2550-2614)
template <typename I, typename O>
int32_t cast_numeric(const orc_view* in, int32_t to_type, bool safe, orc_out* out) {
  int64_t len = in->length;
  const I* iv = (const I*)in->values;
  O* ov = (O*)xalloc((size_t)len * sizeof(O));
  out->type = to_type;
  out->length = len;
  out->values = ov;
  out->values_bytes = len * (int64_t)sizeof(O);
  if (safe) {
    // numeric_cast -> PrimitiveArray::unary_opt (primitive_array.rs:1065-1102):
    // always a null buffer; failed conversions become null
    uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
    if (in->validity) copy_bits(nb, 0, in->validity, in->validity_bit_offset, len);
    else for (int64_t i = 0; i < len; ++i) set_bit(nb, i);
    int64_t nulls = in->validity ? resolve_nulls(in) : 0;
    for (int64_t i = 0; i < len; ++i) {
      if (!get_bit(nb, i)) continue;
      O o;
      if (num_cast_any<I, O>(iv[i], &o)) ov[i] = o;
      else {
        nulls += 1;
        nb[i >> 3] &= (uint8_t)~(1u << (i & 7));
      }
    }
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = nulls;
    return ORC_OK;
  }
  // try_numeric_cast -> try_unary (primitive_array.rs:990-1016): nulls cloned
  uint8_t* nb = nulls_clone(in, len);
  for (int64_t i = 0; i < len; ++i) {
    if (nb && !get_bit(nb, i)) continue;
    O o;
    if (!num_cast_any<I, O>(iv[i], &o)) {
      free(nb);
      orc_release(out);
      return fail(ORC_CAST_ERROR, "Can't cast value %s to type %s", dbg_num(iv[i]).c_str(), type_name(to_type));
    }
    ov[i] = o;
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  }
  return ORC_OK;
}

template <typename I>
int32_t cast_from(const orc_view* in, int32_t to, bool safe, orc_out* out) {
  switch (to) {
    case ORC_INT8: return cast_numeric<I, int8_t>(in, to, safe, out);
    case ORC_INT16: return cast_numeric<I, int16_t>(in, to, safe, out);
    case ORC_INT32: return cast_numeric<I, int32_t>(in, to, safe, out);
    case ORC_INT64: return cast_numeric<I, int64_t>(in, to, safe, out);
    case ORC_UINT8: return cast_numeric<I, uint8_t>(in, to, safe, out);
    case ORC_UINT16: return cast_numeric<I, uint16_t>(in, to, safe, out);
    case ORC_UINT32: return cast_numeric<I, uint32_t>(in, to, safe, out);
    case ORC_UINT64: return cast_numeric<I, uint64_t>(in, to, safe, out);
    case ORC_FLOAT16: return cast_numeric<I, F16>(in, to, safe, out);
    case ORC_FLOAT32: return cast_numeric<I, float>(in, to, safe, out);
    case ORC_FLOAT64: return cast_numeric<I, double>(in, to, safe, out);
  }
  return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
}

// ------------------------------------------------- float -> shortest decimal
// The reference formats floats with the third-party `ryu` crate (1.0.23;
// arrow-cast/src/display.rs:711-723), which is not vendored.  Shortest
// round-trip digits come from libstdc++'s std::to_chars (itself a Ryu
// implementation, independent of the device code); the layout below restates
// ryu's `pretty` module: format64 / format32.
template <typename F>
int format_float(F v, char* out, int kk_max, int kk_min_small) {
  if (v != v) { memcpy(out, "NaN", 3); return 3; }
  if (std::isinf(v)) {
    if (v < 0) { memcpy(out, "-inf", 4); return 4; }
    memcpy(out, "inf", 3);
    return 3;
  }
  int idx = 0;
  if (std::signbit(v)) out[idx++] = '-';
  if (v == 0) { memcpy(out + idx, "0.0", 3); return idx + 3; }
  char sci[64];
  auto res = std::to_chars(sci, sci + sizeof sci, std::fabs(v), std::chars_format::scientific);
  *res.ptr = 0;
  // parse "d.ddddde[+-]xx"
  char digits[32];
  int nd = 0;
  const char* p = sci;
  for (; *p && *p != 'e'; ++p) if (*p != '.') digits[nd++] = *p;
  int exp10 = atoi(p + 1);       // value = d.ddd * 10^exp10
  while (nd > 1 && digits[nd - 1] == '0') --nd;  // (to_chars never emits them, defensive)
  int k = exp10 - (nd - 1);      // value = digits * 10^k
  int kk = nd + k;               // 10^(kk-1) <= v < 10^kk
  if (0 <= k && kk <= kk_max) {  // 1234e7 -> 12340000000.0
    memcpy(out + idx, digits, nd);
    for (int i = nd; i < kk; ++i) out[idx + i] = '0';
    out[idx + kk] = '.';
    out[idx + kk + 1] = '0';
    return idx + kk + 2;
  } else if (0 < kk && kk <= kk_max) {  // 1234e-2 -> 12.34
    memcpy(out + idx, digits, kk);
    out[idx + kk] = '.';
    memcpy(out + idx + kk + 1, digits + kk, nd - kk);
    return idx + nd + 1;
  } else if (kk_min_small < kk && kk <= 0) {  // 1234e-6 -> 0.001234
    out[idx] = '0';
    out[idx + 1] = '.';
    int offset = 2 - kk;
    for (int i = 2; i < offset; ++i) out[idx + i] = '0';
    memcpy(out + idx + offset, digits, nd);
    return idx + nd + offset;
  } else if (nd == 1) {  // 1e30
    out[idx] = digits[0];
    out[idx + 1] = 'e';
    return idx + 2 + snprintf(out + idx + 2, 8, "%d", kk - 1);
  } else {  // 1234e30 -> 1.234e33
    out[idx] = digits[0];
    out[idx + 1] = '.';
    memcpy(out + idx + 2, digits + 1, nd - 1);
    out[idx + nd + 1] = 'e';
    return idx + nd + 2 + snprintf(out + idx + nd + 2, 8, "%d", kk - 1);
  }
}

template <typename T>
int format_value(T v, char* buf) {
  if constexpr (std::is_same<T, double>::value) return orc_format_f64(v, buf);
  else if constexpr (std::is_same<T, float>::value) return orc_format_f32(v, buf);
  else if constexpr (std::is_signed<T>::value) return snprintf(buf, 32, "%lld", (long long)v);  // lexical_core::write (display.rs:694-707)
  else return snprintf(buf, 32, "%llu", (unsigned long long)v);
}

// value_to_string (arrow-cast/src/cast/string.rs:21-39) into GenericStringBuilder
template <typename T, typename OFF>
int32_t cast_to_string(const orc_view* in, int32_t to_type, orc_out* out) {
  int64_t len = in->length;
  const T* iv = (const T*)in->values;
  OFF* offs = (OFF*)xalloc((size_t)(len + 1) * sizeof(OFF));
  std::vector<char> bytes;
  bytes.reserve((size_t)len * 8);
  bool any_null = false;
  uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
  char buf[48];
  offs[0] = 0;
  for (int64_t i = 0; i < len; ++i) {
    bool null = in->validity && !get_bit(in->validity, in->validity_bit_offset + i);
    if (null) any_null = true;
    else {
      int n = format_value<T>(iv[i], buf);
      bytes.insert(bytes.end(), buf, buf + n);
      set_bit(nb, i);
    }
    // OffsetSize::from_usize(..).expect("byte array offset overflow")
    // (arrow-array/src/builder/generic_bytes_builder.rs:86-87)
    if (sizeof(OFF) == 4 && bytes.size() > (size_t)INT32_MAX) {
      free(offs);
      free(nb);
      return fail(ORC_OFFSET_OVERFLOW, "byte array offset overflow");
    }
    offs[i + 1] = (OFF)bytes.size();
  }
  out->type = to_type;
  out->length = len;
  out->offsets = offs;
  out->offsets_bytes = (len + 1) * (int64_t)sizeof(OFF);
  out->values = xalloc(bytes.size());
  memcpy(out->values, bytes.data(), bytes.size());
  out->values_bytes = (int64_t)bytes.size();
  if (any_null) {  // NullBufferBuilder materialises only once a null is appended
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  } else free(nb);
  return ORC_OK;
}

template <typename OFF>
int32_t cast_to_string_dispatch(const orc_view* in, int32_t to, orc_out* out) {
  switch (in->type) {
    case ORC_INT8: return cast_to_string<int8_t, OFF>(in, to, out);
    case ORC_INT16: return cast_to_string<int16_t, OFF>(in, to, out);
    case ORC_INT32: return cast_to_string<int32_t, OFF>(in, to, out);
    case ORC_INT64: return cast_to_string<int64_t, OFF>(in, to, out);
    case ORC_UINT8: return cast_to_string<uint8_t, OFF>(in, to, out);
    case ORC_UINT16: return cast_to_string<uint16_t, OFF>(in, to, out);
    case ORC_UINT32: return cast_to_string<uint32_t, OFF>(in, to, out);
    case ORC_UINT64: return cast_to_string<uint64_t, OFF>(in, to, out);
    case ORC_FLOAT32: return cast_to_string<float, OFF>(in, to, out);
    case ORC_FLOAT64: return cast_to_string<double, OFF>(in, to, out);
  }
  return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
}

inline uint64_t splitmix64(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// neg / neg_wrapping (arrow-arith/src/numeric.rs:103-186)
template <typename T>
int32_t neg_typed(const orc_view* v, bool wrapping, orc_out* out) {
  int64_t len = v->length;
  const T* iv = (const T*)v->values;
  T* ov = (T*)xalloc((size_t)len * sizeof(T));
  out->type = v->type;
  out->length = len;
  out->values = ov;
  out->values_bytes = len * (int64_t)sizeof(T);
  uint8_t* nb = nulls_clone(v, len);
  for (int64_t i = 0; i < len; ++i) {
    if constexpr (std::is_same<T, F16>::value) ov[i] = F16{(uint16_t)(iv[i].bits ^ 0x8000u)};  // Neg for f16
    else if constexpr (std::is_floating_point<T>::value) ov[i] = -iv[i];
    else if (wrapping) ov[i] = (T)(0 - (typename std::make_unsigned<T>::type)iv[i]);
    else {
      if (nb && !get_bit(nb, i)) continue;
      if (iv[i] == std::numeric_limits<T>::min()) {
        free(nb);
        std::string s = dbg(iv[i]);
        orc_release(out);
        return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: - %s", s.c_str());
      }
      ov[i] = (T)-iv[i];
    }
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  }
  return ORC_OK;
}

// ---------------------------------------------------------------- aggregate
// arrow-arith/src/aggregate.rs.  Accumulators :51-177, lane kernels :179-297, dispatch :317-361.
enum { G_SUM = 0, G_SUM_CHECKED, G_PRODUCT, G_PRODUCT_CHECKED, G_MIN, G_MAX, G_BIT_AND, G_BIT_OR, G_BIT_XOR };

template <typename T> struct TotalOrderLimits {  // MIN_TOTAL_ORDER / MAX_TOTAL_ORDER (arithmetic.rs:45-56)
  static T min() { return std::numeric_limits<T>::min(); }
  static T max() { return std::numeric_limits<T>::max(); }
};
template <> struct TotalOrderLimits<double> {  // -NaN / +NaN with every payload bit set
  static double from(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
  static double min() { return from(0xFFFFFFFFFFFFFFFFull); }
  static double max() { return from(0x7FFFFFFFFFFFFFFFull); }
};
template <> struct TotalOrderLimits<float> {
  static float from(uint32_t b) { float d; memcpy(&d, &b, 4); return d; }
  static float min() { return from(0xFFFFFFFFu); }
  static float max() { return from(0x7FFFFFFFu); }
};
template <typename T> inline T agg_add(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return a + b;
  else return int_wrapping<T>(OP_ADD_W, a, b);
}
template <typename T> inline T agg_mul(T a, T b) {
  if constexpr (std::is_floating_point<T>::value) return a * b;
  else return int_wrapping<T>(OP_MUL_W, a, b);
}
template <typename T, int KIND> struct Acc {  // NumericAccumulator (:32-42)
  T v;
  Acc() {
    if (KIND == G_SUM) v = T(0);
    else if (KIND == G_PRODUCT) v = T(1);
    else if (KIND == G_MIN) v = TotalOrderLimits<T>::max();
    else v = TotalOrderLimits<T>::min();
  }
  void accumulate(T x) {
    if (KIND == G_SUM) v = agg_add<T>(v, x);
    else if (KIND == G_PRODUCT) v = agg_mul<T>(v, x);
    else if (KIND == G_MIN) v = is_lt<T>(x, v) ? x : v;
    else v = is_lt<T>(v, x) ? x : v;  // is_gt
  }
  void accumulate_nullable(T x, bool valid) {
    if (valid) accumulate(x);  // select(valid, op(acc, x), acc)
  }
  void merge(const Acc& o) { accumulate(o.v); }  // Sum/Product: op with other; Min/Max: accumulate(other)
};

template <typename T, int KIND> T reduce_accumulators(std::vector<Acc<T, KIND>>& acc) {  // :179-197
  size_t len = acc.size();
  while (len >= 2) {
    size_t mid = len / 2;
    for (size_t i = 0; i < mid; ++i) acc[i].merge(acc[mid + i]);
    len /= 2;
  }
  return acc[0].v;
}
template <typename T, int KIND> T aggregate_nonnull_simple(const T* v, int64_t n) {  // :222-231
  Acc<T, KIND> a;
  for (int64_t i = 0; i < n; ++i) a.accumulate(v[i]);
  return a.v;
}
template <typename T, int KIND> T aggregate_nonnull_lanes(const T* v, int64_t n, int lanes) {  // :234-251
  std::vector<Acc<T, KIND>> acc(lanes);
  int64_t full = n / lanes * lanes;
  for (int64_t i = 0; i < full; i += lanes)
    for (int l = 0; l < lanes; ++l) acc[l].accumulate(v[i + l]);
  for (int64_t i = full; i < n; ++i) acc[i - full].accumulate(v[i]);
  return reduce_accumulators<T, KIND>(acc);
}
template <typename T, int KIND>
T aggregate_nullable_lanes(const T* v, int64_t n, const uint8_t* bits, int64_t off, int lanes) {  // :254-297
  // 64-row validity chunks, each cut into `lanes`-wide groups; the tail (< 64 rows) the same way with
  // a final partial group feeding acc[0..rem) — identical to indexing lane = (row % 64) % lanes.
  std::vector<Acc<T, KIND>> acc(lanes);
  for (int64_t i = 0; i < n; ++i) acc[(i % 64) % lanes].accumulate_nullable(v[i], get_bit(bits, off + i));
  return reduce_accumulators<T, KIND>(acc);
}
inline int lanes_for(int vector_bytes, size_t width) {  // the `match` at :329-337 / :345-353
  int q = (int)(vector_bytes / width);
  switch (q) {
    case 64: case 32: case 16: case 8: case 4: case 2: return q;
    default: return 1;
  }
}

template <typename T> void store_scalar(orc_scalar* out, T v) {
  out->is_valid = 1;
  memcpy(out->bytes, &v, sizeof(T));
}

template <typename T, int KIND> int32_t aggregate_numeric(const orc_view* a, int vector_bytes, orc_scalar* out) {
  const int64_t nulls = resolve_nulls(a), n = a->length;
  if (nulls == n) return ORC_OK;  // None (:320-323), also the empty array
  const T* v = (const T*)a->values;
  if (a->validity && nulls > 0) {
    store_scalar<T>(out, aggregate_nullable_lanes<T, KIND>(v, n, a->validity, a->validity_bit_offset,
                                                            lanes_for(vector_bytes, sizeof(T))));
  } else if (std::is_floating_point<T>::value) {
    int lanes = lanes_for(vector_bytes * 2, sizeof(T));  // PREFERRED_VECTOR_SIZE_NON_NULL (:310)
    store_scalar<T>(out, lanes > 1 ? aggregate_nonnull_lanes<T, KIND>(v, n, lanes)
                                   : aggregate_nonnull_simple<T, KIND>(v, n));
  } else {
    store_scalar<T>(out, aggregate_nonnull_simple<T, KIND>(v, n));
  }
  return ORC_OK;
}

// sum_checked / product_checked (:897-933, :963-1001): strictly sequential over the valid slots
template <typename T> int32_t aggregate_checked(int kind, const orc_view* a, orc_scalar* out) {
  const int64_t nulls = resolve_nulls(a), n = a->length;
  if (nulls == n) return ORC_OK;
  const T* v = (const T*)a->values;
  T acc = kind == G_SUM_CHECKED ? T(0) : T(1);
  for (int64_t i = 0; i < n; ++i) {
    if (a->validity && !get_bit(a->validity, a->validity_bit_offset + i)) continue;
    if constexpr (std::is_floating_point<T>::value) {
      acc = kind == G_SUM_CHECKED ? acc + v[i] : acc * v[i];  // float *_checked never fails
    } else {
      T r;
      int32_t st = int_checked<T>(kind == G_SUM_CHECKED ? OP_ADD : OP_MUL, acc, v[i], &r);
      if (st != ORC_OK) return st;
      acc = r;
    }
  }
  store_scalar<T>(out, acc);
  return ORC_OK;
}

// bit_and / bit_or / bit_xor (:776-873)
template <typename T> int32_t aggregate_bits(int kind, const orc_view* a, orc_scalar* out) {
  const int64_t nulls = resolve_nulls(a), n = a->length;
  if (nulls == n) return ORC_OK;
  using U = typename std::make_unsigned<T>::type;
  const U* v = (const U*)a->values;
  U r = kind == G_BIT_AND ? (U)~(U)0 : (U)0;
  for (int64_t i = 0; i < n; ++i) {
    if (a->validity && !get_bit(a->validity, a->validity_bit_offset + i)) continue;
    r = kind == G_BIT_AND ? (U)(r & v[i]) : kind == G_BIT_OR ? (U)(r | v[i]) : (U)(r ^ v[i]);
  }
  store_scalar<U>(out, r);
  return ORC_OK;
}

template <typename T> int32_t aggregate_dispatch(int op, const orc_view* a, int vb, orc_scalar* out) {
  switch (op) {
    case G_SUM: return aggregate_numeric<T, G_SUM>(a, vb, out);
    case G_PRODUCT: return aggregate_numeric<T, G_PRODUCT>(a, vb, out);
    case G_MIN: return aggregate_numeric<T, G_MIN>(a, vb, out);
    case G_MAX: return aggregate_numeric<T, G_MAX>(a, vb, out);
    case G_SUM_CHECKED: case G_PRODUCT_CHECKED: return aggregate_checked<T>(op, a, out);
    default:
      if constexpr (std::is_integral<T>::value) return aggregate_bits<T>(op, a, out);
      return fail(ORC_INVALID_ARGUMENT, "bitwise aggregates need an integer type");
  }
}

// min_boolean / max_boolean (= bool_and / bool_or) (:372-457, :880-889)
int32_t aggregate_boolean(int op, const orc_view* a, orc_scalar* out) {
  const int64_t nulls = resolve_nulls(a), n = a->length;
  if (nulls == n) return ORC_OK;
  const uint8_t* v = (const uint8_t*)a->values;
  bool any_true = false, any_false = false;
  for (int64_t i = 0; i < n; ++i) {
    if (a->validity && !get_bit(a->validity, a->validity_bit_offset + i)) continue;
    if (get_bit(v, a->values_bit_offset + i)) any_true = true;
    else any_false = true;
  }
  if (op == G_MIN) store_scalar<uint8_t>(out, any_false ? 0 : 1);
  else if (op == G_MAX) store_scalar<uint8_t>(out, any_true ? 1 : 0);
  else return fail(ORC_INVALID_ARGUMENT, "only min/max (bool_and/bool_or) aggregate a BooleanArray");
  return ORC_OK;
}

template <typename T>
void sort_valid_typed(const orc_view* a, std::vector<uint32_t>& valid, bool desc) {
  const T* v = (const T*)a->values;
  if (!desc) std::stable_sort(valid.begin(), valid.end(), [v](uint32_t x, uint32_t y) { return is_lt<T>(v[x], v[y]); });
  else std::stable_sort(valid.begin(), valid.end(), [v](uint32_t x, uint32_t y) { return is_lt<T>(v[y], v[x]); });
}

// Boolean <-> numeric (arrow-cast/src/cast/mod.rs:1243-1290): numeric_to_bool_cast :2661 (value != default; a null
// buffer only if a null was appended), bool_to_numeric_cast :2704 (true -> 1, false -> 0, null slots default;
// from_trusted_len_iter always builds a null buffer)
template <typename T> int32_t cast_num_to_bool(const orc_view* in, orc_out* out) {
  const int64_t n = in->length;
  out->type = ORC_BOOL;
  out->length = n;
  const T* v = (const T*)in->values;
  uint8_t* vals = (uint8_t*)xalloc(bitmap_bytes(n));
  const bool has_nulls = in->validity && resolve_nulls(in) > 0;
  uint8_t* nb = has_nulls ? (uint8_t*)xalloc(bitmap_bytes(n)) : nullptr;
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (has_nulls && !get_bit(in->validity, in->validity_bit_offset + i)) {
      ++nulls;
      continue;
    }
    if (nb) set_bit(nb, i);
    if (v[i] != T{}) set_bit(vals, i);
  }
  out->values = vals;
  out->values_bytes = (int64_t)bitmap_bytes(n);
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(n);
    out->null_count = nulls;
  }
  return ORC_OK;
}
template <typename T> int32_t cast_bool_to_num(const orc_view* in, int32_t to, orc_out* out) {
  const int64_t n = in->length;
  out->type = to;
  out->length = n;
  T* o = (T*)xalloc((size_t)std::max<int64_t>(n, 1) * sizeof(T));
  uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(n));
  int64_t nulls = 0;
  const uint8_t* b = (const uint8_t*)in->values;
  for (int64_t i = 0; i < n; ++i) {
    if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) {
      o[i] = T{};
      ++nulls;
      continue;
    }
    set_bit(nb, i);
    o[i] = get_bit(b, in->values_bit_offset + i) ? (T)1 : T{};
  }
  out->values = o;
  out->values_bytes = std::max<int64_t>(n, 1) * (int64_t)sizeof(T);
  out->validity = nb;
  out->validity_bytes = (int64_t)bitmap_bytes(n);
  out->null_count = nulls;
  return ORC_OK;
}

// arrow_arith::bitwise (bitwise.rs:42-135): `binary` (all slots, nulls = presence-based union) / `unary` (nulls cloned)
template <typename T>
int32_t bitwise_typed(int op, const orc_view* l, bool l_s, const orc_view* r, bool r_s, orc_out* out) {
  using U = typename std::make_unsigned<T>::type;
  auto f = [op](T a, T b) -> T {
    const unsigned sh = (unsigned)(U)b & (sizeof(T) * 8 - 1);  // wrapping_shl/shr(b as usize as u32)
    switch (op) {
      case 8: return (T)(a & b);
      case 9: return (T)(a | b);
      case 10: return (T)(a ^ b);
      case 11: return (T)((U)a << sh);
      case 12: return (T)(a >> sh);
      case 13: return (T)(a & ~b);
      default: return (T)~a;
    }
  };
  out->type = l->type;
  const T* lv = (const T*)l->values;
  const T* rv = (const T*)r->values;
  if (op == 14 || l_s != r_s) {  // unary / `_scalar` forms
    const orc_view* arr = (op == 14 || !l_s) ? l : r;
    const int64_t len = arr->length;
    out->length = len;
    T* ov = (T*)xalloc((size_t)std::max<int64_t>(len, 1) * sizeof(T));
    out->values = ov;
    out->values_bytes = std::max<int64_t>(len, 1) * (int64_t)sizeof(T);
    for (int64_t i = 0; i < len; ++i) ov[i] = f(l_s ? lv[0] : lv[i], (op == 14) ? T{} : (r_s ? rv[0] : rv[i]));
    uint8_t* nb = nulls_clone(arr, len);
    if (nb) {
      out->validity = nb;
      out->validity_bytes = (int64_t)bitmap_bytes(len);
      out->null_count = len - count_set_bits(nb, 0, len);
    }
    return ORC_OK;
  }
  if (l->length != r->length) return fail(ORC_COMPUTE_ERROR, "Cannot perform binary operation on arrays of different length");
  const int64_t len = l->length;
  out->length = len;
  if (len == 0) return ORC_OK;
  T* ov = (T*)xalloc((size_t)len * sizeof(T));
  out->values = ov;
  out->values_bytes = len * (int64_t)sizeof(T);
  for (int64_t i = 0; i < len; ++i) ov[i] = f(lv[i], rv[i]);
  int64_t nc = 0;
  uint8_t* nb = nulls_union(l, r, len, &nc);
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = nc;
  }
  return ORC_OK;
}

}  // namespace

// =================================================================== exports
namespace {

// ---- Utf8 / LargeUtf8 -> numeric (arrow-cast/src/cast/string.rs:66-120 parse_string -> Parser::parse, parse.rs:446-528)
bool is_ascii_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == 0x0C || c == '\r'; }  // u8::is_ascii_whitespace
bool is_dec_digit(uint8_t c) { return c >= '0' && c <= '9'; }

// atoi::FromRadix10SignedChecked (atoi 3.1.0): optional sign, then every following digit with checked arithmetic;
// returns the number of bytes used and whether the value survived
template <typename T>
size_t atoi_signed_checked(const uint8_t* t, size_t n, bool* some, T* value) {
  size_t i = 0;
  bool neg = false;
  if (n > 0 && (t[0] == '+' || t[0] == '-')) {
    neg = t[0] == '-';
    i = 1;
  }
  T acc = 0;
  bool ok = true;
  for (; i < n && is_dec_digit(t[i]); ++i) {
    const T d = (T)(t[i] - '0');
    if (ok && __builtin_mul_overflow(acc, (T)10, &acc)) ok = false;
    if (ok && (neg ? __builtin_sub_overflow(acc, d, &acc) : __builtin_add_overflow(acc, d, &acc))) ok = false;
  }
  *some = ok;
  *value = acc;
  return i;
}

// parser_primitive! (parse.rs:492-516)
template <typename T>
bool parse_native_int(const uint8_t* raw, size_t n, T* out) {
  auto last_is_digit = [&]() { return n > 0 && is_dec_digit(raw[n - 1]); };
  if (!last_is_digit()) {
    while (n > 0 && is_ascii_ws(raw[n - 1])) --n;  // trim_ascii_end
    if (!last_is_digit()) return false;
  }
  bool some;
  T v;
  if (atoi_signed_checked<T>(raw, n, &some, &v) == n && some) {
    *out = v;
    return true;
  }
  while (n > 0 && is_ascii_ws(raw[0])) {  // trim_ascii_start
    ++raw;
    --n;
  }
  if (atoi_signed_checked<T>(raw, n, &some, &v) == n && some) {
    *out = v;
    return true;
  }
  return false;
}

// lexical_core::parse::<f32 | f64> (lexical-core 1.0.6, standard format; sources not under /root/reference — the
// grammar is restated from its documentation): [+-] (digits [. digits*] | . digits) [(e|E) [+-] digits], or
// case-insensitive nan / inf / infinity after the sign; the whole text must be consumed.  The VALUE is the correctly
// rounded float, which glibc's strtod / strtof also produce: they do the arithmetic here.
template <typename F>
bool lexical_parse_float(const uint8_t* t, size_t n, F* out) {
  size_t i = 0;
  if (n == 0) return false;
  bool neg = false;
  if (t[0] == '+' || t[0] == '-') {
    neg = t[0] == '-';
    i = 1;
  }
  const size_t body = i;
  size_t digits = 0;
  while (i < n && is_dec_digit(t[i])) ++i, ++digits;
  if (i < n && t[i] == '.') {
    ++i;
    while (i < n && is_dec_digit(t[i])) ++i, ++digits;
  }
  if (digits == 0) {
    std::string low;
    for (size_t k = body; k < n; ++k) low += (char)tolower(t[k]);
    F v;
    if (low == "nan") v = std::numeric_limits<F>::quiet_NaN();
    else if (low == "inf" || low == "infinity") v = std::numeric_limits<F>::infinity();
    else return false;
    *out = neg ? -v : v;
    return true;
  }
  if (i < n && (t[i] == 'e' || t[i] == 'E')) {
    ++i;
    if (i < n && (t[i] == '+' || t[i] == '-')) ++i;
    size_t ed = 0;
    while (i < n && is_dec_digit(t[i])) ++i, ++ed;
    if (ed == 0) return false;
  }
  if (i != n) return false;
  const std::string z((const char*)t, n);
  *out = sizeof(F) == 8 ? (F)strtod(z.c_str(), nullptr) : (F)strtof(z.c_str(), nullptr);
  return true;
}

// Float32Type / Float64Type::parse (parse.rs:458-475): the text, then the text trimmed on both sides
template <typename F>
bool parse_native_float(const uint8_t* raw, size_t n, F* out) {
  if (lexical_parse_float<F>(raw, n, out)) return true;
  while (n > 0 && is_ascii_ws(raw[0])) ++raw, --n;
  while (n > 0 && is_ascii_ws(raw[n - 1])) --n;
  return lexical_parse_float<F>(raw, n, out);
}

template <typename T>
bool parse_native(const uint8_t* raw, size_t n, T* out) {
  if constexpr (std::is_floating_point<T>::value) return parse_native_float<T>(raw, n, out);
  else return parse_native_int<T>(raw, n, out);
}

// parse_string_iter (string.rs:87-120)
template <typename T>
int32_t parse_string(const orc_view* in, int32_t to, int32_t safe, orc_out* out) {
  const int64_t len = in->length;
  const bool large = in->type == ORC_LARGE_UTF8;
  out->type = to;
  out->length = len;
  if (len == 0) return ORC_OK;
  auto off = [&](int64_t i) -> int64_t { return large ? ((const int64_t*)in->offsets)[i] : (int64_t)((const int32_t*)in->offsets)[i]; };
  T* vals = (T*)xalloc((size_t)len * sizeof(T));
  uint8_t* nb = safe ? (uint8_t*)xalloc(bitmap_bytes(len)) : nullptr;
  for (int64_t i = 0; i < len; ++i) {
    const bool valid = !in->validity || get_bit(in->validity, in->validity_bit_offset + i);
    T v{};
    if (valid) {
      const int64_t a = off(i), b = off(i + 1);
      const bool ok = parse_native<T>((const uint8_t*)in->values + a, (size_t)(b - a), &v);
      if (!ok) {
        v = T{};
        if (!safe) {
          const std::string text((const char*)in->values + a, (size_t)(b - a));
          free(vals);
          return fail(ORC_CAST_ERROR, "Cannot cast string '%s' to value of %s type", text.c_str(), type_name(to));
        }
      } else if (nb) {
        set_bit(nb, i);
      }
    }
    vals[i] = v;
  }
  out->values = vals;
  out->values_bytes = len * (int64_t)sizeof(T);
  if (safe) {  // from_trusted_len_iter: always a null buffer
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  } else if (in->validity) {  // try_new(values, nulls().cloned())
    out->validity = nulls_clone(in, len);
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(out->validity, 0, len);
  }
  return ORC_OK;
}


}  // namespace

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

void orc_release(orc_out* out) {
  if (!out) return;
  if (!(out->flags & 1)) {
    if (!(out->flags & 2)) {
      free(out->values);
      free(out->offsets);
    }
    free(out->validity);
  }
  memset(out, 0, sizeof *out);
}

int64_t orc_count_set_bits(const uint8_t* bits, int64_t off, int64_t len) {
  return count_set_bits(bits, off, len);
}
int64_t orc_set_slices(const uint8_t* bits, int64_t off, int64_t len, int64_t* out, int64_t cap) {
  BitSliceIterator it(bits, off, len);
  int64_t s, e, n = 0;
  while (it.next(&s, &e)) {
    if (n < cap) {
      out[2 * n] = s;
      out[2 * n + 1] = e;
    }
    ++n;
  }
  return n;
}
int64_t orc_set_indices(const uint8_t* bits, int64_t off, int64_t len, int64_t* out, int64_t cap) {
  BitIndexIterator it(bits, off, len);
  int64_t i, n = 0;
  while (it.next(&i)) {
    if (n < cap) out[n] = i;
    ++n;
  }
  return n;
}

int32_t orc_filter(const orc_view* values, const orc_view* predicate, orc_out* out) {
  return filter_impl(values, predicate, out);
}

// IterationStrategy::default_strategy (filter.rs:346-364) for this predicate: 0 None, 1 All, 2 SlicesIterator, 3 IndexIterator
int32_t orc_filter_strategy(const orc_view* predicate) {
  Predicate p;
  build_predicate(predicate, &p);
  return (int32_t)p.strategy;
}

int32_t orc_take(const orc_view* values, const orc_view* indices, int32_t cb, orc_out* out) {
  out_init(out);
  if (type_width(values->type) < 0 && values->type != ORC_UTF8 && values->type != ORC_LARGE_UTF8)
    return fail(ORC_NOT_YET_IMPLEMENTED, "take not supported for type %s", type_name(values->type));
  switch (indices->type) {  // downcast_integer_array! (take.rs:95-105)
    case ORC_INT8: return take_typed<int8_t>(values, indices, cb, out);
    case ORC_UINT8: return take_typed<uint8_t>(values, indices, cb, out);
    case ORC_INT16: return take_typed<int16_t>(values, indices, cb, out);
    case ORC_UINT16: return take_typed<uint16_t>(values, indices, cb, out);
    case ORC_INT32: return take_typed<int32_t>(values, indices, cb, out);
    case ORC_UINT32: return take_typed<uint32_t>(values, indices, cb, out);
    case ORC_INT64: return take_typed<int64_t>(values, indices, cb, out);
    case ORC_UINT64: return take_typed<uint64_t>(values, indices, cb, out);
  }
  return fail(ORC_INVALID_ARGUMENT, "Take only supported for integers, got %s", type_name(indices->type));
}

int32_t orc_arith(int32_t op, const orc_view* l, int32_t l_s, const orc_view* r, int32_t r_s, orc_out* out) {
  out_init(out);
  if (l->type != r->type || !(is_integer(l->type) || l->type == ORC_FLOAT16 || l->type == ORC_FLOAT32 || l->type == ORC_FLOAT64))
    return fail(ORC_INVALID_ARGUMENT, "Invalid arithmetic operation: %s %s %s", type_name(l->type),
                op_sym(op), type_name(r->type));  // numeric.rs:270-272
  switch (l->type) {
    case ORC_INT8: return arith_typed<int8_t>(op, l, l_s, r, r_s, out);
    case ORC_INT16: return arith_typed<int16_t>(op, l, l_s, r, r_s, out);
    case ORC_INT32: return arith_typed<int32_t>(op, l, l_s, r, r_s, out);
    case ORC_INT64: return arith_typed<int64_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT8: return arith_typed<uint8_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT16: return arith_typed<uint16_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT32: return arith_typed<uint32_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT64: return arith_typed<uint64_t>(op, l, l_s, r, r_s, out);
    case ORC_FLOAT16: return arith_typed<F16>(op, l, l_s, r, r_s, out);  // numeric.rs:240
    case ORC_FLOAT32: return arith_typed<float>(op, l, l_s, r, r_s, out);
    default: return arith_typed<double>(op, l, l_s, r, r_s, out);
  }
}

int32_t orc_neg(const orc_view* v, int32_t wrapping, orc_out* out) {
  out_init(out);
  switch (v->type) {
    case ORC_INT8: return neg_typed<int8_t>(v, wrapping, out);
    case ORC_INT16: return neg_typed<int16_t>(v, wrapping, out);
    case ORC_INT32: return neg_typed<int32_t>(v, wrapping, out);
    case ORC_INT64: return neg_typed<int64_t>(v, wrapping, out);
    case ORC_FLOAT16: return neg_typed<F16>(v, wrapping, out);
    case ORC_FLOAT32: return neg_typed<float>(v, wrapping, out);
    case ORC_FLOAT64: return neg_typed<double>(v, wrapping, out);
    case ORC_UINT8: if (wrapping) return neg_typed<uint8_t>(v, true, out); break;
    case ORC_UINT16: if (wrapping) return neg_typed<uint16_t>(v, true, out); break;
    case ORC_UINT32: if (wrapping) return neg_typed<uint32_t>(v, true, out); break;
    case ORC_UINT64: if (wrapping) return neg_typed<uint64_t>(v, true, out); break;
  }
  return fail(ORC_INVALID_ARGUMENT, "Invalid arithmetic operation: !%s", type_name(v->type));
}

// compare_op (arrow-ord/src/cmp.rs:220-382)
int32_t orc_compare(int32_t op, const orc_view* l, int32_t l_s, const orc_view* r, int32_t r_s, orc_out* out) {
  out_init(out);
  if (l->length != r->length && !l_s && !r_s)
    return fail(ORC_INVALID_ARGUMENT, "Cannot compare arrays of different lengths, got %lld vs %lld",
                (long long)l->length, (long long)r->length);
  int64_t len = l_s ? r->length : l->length;
  if (l->type != r->type)
    return fail(ORC_INVALID_ARGUMENT, "Invalid comparison operation: %s %s %s", type_name(l->type),
                cmp_sym(op), type_name(r->type));
  out->type = ORC_BOOL;
  out->length = len;
  auto values = [&]() -> uint8_t* {
    if (l->length == 0 || r->length == 0) return (uint8_t*)xalloc(bitmap_bytes(len));  // apply :445-447
    switch (l->type) {
      case ORC_BOOL: return cmp_values_bool(op, l, l_s, r, r_s, len);
      case ORC_INT8: return cmp_values<int8_t>(op, l, l_s, r, r_s, len);
      case ORC_INT16: return cmp_values<int16_t>(op, l, l_s, r, r_s, len);
      case ORC_INT32: return cmp_values<int32_t>(op, l, l_s, r, r_s, len);
      case ORC_INT64: return cmp_values<int64_t>(op, l, l_s, r, r_s, len);
      case ORC_UINT8: return cmp_values<uint8_t>(op, l, l_s, r, r_s, len);
      case ORC_UINT16: return cmp_values<uint16_t>(op, l, l_s, r, r_s, len);
      case ORC_UINT32: return cmp_values<uint32_t>(op, l, l_s, r, r_s, len);
      case ORC_UINT64: return cmp_values<uint64_t>(op, l, l_s, r, r_s, len);
      case ORC_FLOAT16: return cmp_values<F16>(op, l, l_s, r, r_s, len);
      case ORC_FLOAT32: return cmp_values<float>(op, l, l_s, r, r_s, len);
      case ORC_FLOAT64: return cmp_values<double>(op, l, l_s, r, r_s, len);
      case ORC_UTF8: case ORC_LARGE_UTF8: return cmp_values_bytes(op, l, l_s, r, r_s, len);
      case ORC_FIXED16: return cmp_values_i128(op, l, l_s, r, r_s, len);
    }
    return nullptr;
  };
  const bool is_bytes = l->type == ORC_UTF8 || l->type == ORC_LARGE_UTF8;
  if ((type_width(l->type) < 0 && !is_bytes) || l->type == ORC_FIXED32)
    return fail(ORC_NOT_YET_IMPLEMENTED, "comparison not supported for type %s", type_name(l->type));
  size_t bytes = bitmap_bytes(len);
  // nulls filtered by null_count > 0 (:345-346)
  bool ln = l->validity && resolve_nulls(l) > 0;
  bool rn = r->validity && resolve_nulls(r) > 0;
  auto lbit = [&](int64_t i) { return get_bit(l->validity, l->validity_bit_offset + (l_s ? 0 : i)); };
  auto rbit = [&](int64_t i) { return get_bit(r->validity, r->validity_bit_offset + (r_s ? 0 : i)); };
  auto set_values = [&](uint8_t* v) {
    out->values = v;
    out->values_bytes = (int64_t)bytes;
  };
  auto all_null = [&]() {  // BooleanArray::new_null(len)
    set_values((uint8_t*)xalloc(bytes));
    out->validity = (uint8_t*)xalloc(bytes);
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len;
  };
  if (ln && rn && (l_s == r_s)) {
    uint8_t* v = values();
    if (op == C_DISTINCT || op == C_NOT_DISTINCT) {
      uint8_t* o = (uint8_t*)xalloc(bytes);
      for (int64_t i = 0; i < len; ++i) {
        bool a = lbit(i), b = rbit(i), n = get_bit(v, i);
        bool res = op == C_DISTINCT ? ((a ^ b) | (a & b & n)) : ((!(a | b)) | (a & b & n));
        if (res) set_bit(o, i);
      }
      free(v);
      set_values(o);
    } else {
      set_values(v);
      uint8_t* nb = (uint8_t*)xalloc(bytes);
      for (int64_t i = 0; i < len; ++i) if (lbit(i) && rbit(i)) set_bit(nb, i);
      out->validity = nb;
      out->validity_bytes = (int64_t)bytes;
      out->null_count = len - count_set_bits(nb, 0, len);
    }
  } else if (ln && rn) {  // scalar is null, other side non-scalar and nullable (:349-356)
    const orc_view* a = l_s ? r : l;
    if (op == C_DISTINCT || op == C_NOT_DISTINCT) {
      uint8_t* o = (uint8_t*)xalloc(bytes);
      for (int64_t i = 0; i < len; ++i) {
        bool va = get_bit(a->validity, a->validity_bit_offset + i);
        if (op == C_DISTINCT ? va : !va) set_bit(o, i);
      }
      set_values(o);
    } else all_null();
  } else if (ln || rn) {  // only one side nullable (:357-378)
    const orc_view* nside = ln ? l : r;
    bool is_scalar = ln ? l_s : r_s;
    if (is_scalar) {
      if (op == C_DISTINCT) {
        uint8_t* o = (uint8_t*)xalloc(bytes);
        for (int64_t i = 0; i < len; ++i) set_bit(o, i);
        set_values(o);
      } else if (op == C_NOT_DISTINCT) set_values((uint8_t*)xalloc(bytes));
      else all_null();
    } else {
      uint8_t* v = values();
      auto nbit = [&](int64_t i) { return get_bit(nside->validity, nside->validity_bit_offset + i); };
      if (op == C_DISTINCT || op == C_NOT_DISTINCT) {
        uint8_t* o = (uint8_t*)xalloc(bytes);
        for (int64_t i = 0; i < len; ++i) {
          bool res = op == C_DISTINCT ? (!nbit(i) | get_bit(v, i)) : (nbit(i) & get_bit(v, i));
          if (res) set_bit(o, i);
        }
        free(v);
        set_values(o);
      } else {
        set_values(v);
        uint8_t* nb = (uint8_t*)xalloc(bytes);
        for (int64_t i = 0; i < len; ++i) if (nbit(i)) set_bit(nb, i);
        out->validity = nb;
        out->validity_bytes = (int64_t)bytes;
        out->null_count = len - count_set_bits(nb, 0, len);
      }
    }
  } else {
    set_values(values());
  }
  return ORC_OK;
}

// arrow_arith::boolean (arrow-arith/src/boolean.rs): and/or/and_not :256-300 via
// binary_boolean_kernel :224 (nulls = NullBuffer::union), and_kleene :60-151, or_kleene :156-222
int32_t orc_boolean_binary(int32_t op, const orc_view* l, const orc_view* r, orc_out* out) {
  out_init(out);
  if (l->type != ORC_BOOL || r->type != ORC_BOOL)
    return fail(ORC_INVALID_ARGUMENT, "boolean kernels need Boolean inputs, got %s and %s", type_name(l->type), type_name(r->type));
  if (l->length != r->length)
    return fail(ORC_COMPUTE_ERROR, "Cannot perform bitwise operation on arrays of different length");
  int64_t len = l->length;
  out->type = ORC_BOOL;
  out->length = len;
  if (len == 0) return ORC_OK;
  size_t bytes = bitmap_bytes(len);
  uint8_t* vals = (uint8_t*)xalloc(bytes);
  const uint8_t* lb = (const uint8_t*)l->values;
  const uint8_t* rb = (const uint8_t*)r->values;
  auto LV = [&](int64_t i) { return get_bit(lb, l->values_bit_offset + i); };
  auto RV = [&](int64_t i) { return get_bit(rb, r->values_bit_offset + i); };
  auto LN = [&](int64_t i) { return get_bit(l->validity, l->validity_bit_offset + i); };
  auto RN = [&](int64_t i) { return get_bit(r->validity, r->validity_bit_offset + i); };
  for (int64_t i = 0; i < len; ++i) {
    bool a = LV(i), b = RV(i), v;
    switch (op) {
      case 0: case 3: v = a && b; break;
      case 2: v = a && !b; break;
      default: v = a || b; break;
    }
    if (v) set_bit(vals, i);
  }
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (!l->validity && !r->validity) return ORC_OK;
  uint8_t* nb = (uint8_t*)xalloc(bytes);
  for (int64_t i = 0; i < len; ++i) {
    bool n;
    if (op == 3 || op == 4) {
      bool is_and = op == 3;
      if (l->validity && r->validity) {
        bool a = LN(i), b = LV(i), c = RN(i), d = RV(i);
        n = is_and ? ((a | (c & !d)) & (c | (a & !b))) : ((a | (c & d)) & (c | (a & b)));
      } else if (l->validity) n = is_and ? (LN(i) | !RV(i)) : (LN(i) | RV(i));
      else n = is_and ? (RN(i) | !LV(i)) : (RN(i) | LV(i));
    } else {
      n = (!l->validity || LN(i)) && (!r->validity || RN(i));
    }
    if (n) set_bit(nb, i);
  }
  out->validity = nb;
  out->validity_bytes = (int64_t)bytes;
  out->null_count = len - count_set_bits(nb, 0, len);
  return ORC_OK;
}

// not :310, is_null :327, is_not_null :347
int32_t orc_boolean_unary(int32_t op, const orc_view* v, orc_out* out) {
  out_init(out);
  if (op == 10 && v->type != ORC_BOOL)
    return fail(ORC_INVALID_ARGUMENT, "not() needs a Boolean input, got %s", type_name(v->type));
  int64_t len = v->length;
  out->type = ORC_BOOL;
  out->length = len;
  if (len == 0) return ORC_OK;
  size_t bytes = bitmap_bytes(len);
  uint8_t* vals = (uint8_t*)xalloc(bytes);
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  for (int64_t i = 0; i < len; ++i) {
    bool b;
    if (op == 10) b = !get_bit((const uint8_t*)v->values, v->values_bit_offset + i);
    else {
      bool valid = !v->validity || get_bit(v->validity, v->validity_bit_offset + i);
      b = op == 11 ? !valid : valid;
    }
    if (b) set_bit(vals, i);
  }
  if (op == 10 && v->validity) {
    out->validity = nulls_clone(v, len);
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len - count_set_bits(out->validity, 0, len);
  }
  return ORC_OK;
}

// ------------------------------------------------------------------ string predicates (arrow-string)
// like / nlike / starts_with / ends_with / contains against a scalar pattern: like_op (like.rs:218) -> op_scalar
// (:349) -> Predicate::{like, contains, StartsWith, EndsWith} (predicate.rs:44-120).  The reference hands
// everything that is not Eq / StartsWith / EndsWith / Contains to the `regex` crate (regex 1.x, not under
// /root/reference); the only constructs regex_like (predicate.rs:246-306) ever emits are literal characters,
// `.` and `.*` with dot-matches-newline, and the two anchors, so they are matched here by a backtracking
// matcher over code points — deliberately NOT the two-pointer walk the device uses.
namespace {
struct LikeRegex {
  struct Tok { int kind; uint32_t cp; };  // 0 literal, 1 `.`, 2 `.*`
  std::vector<Tok> toks;
  bool anchored_start = true, anchored_end = true;
};

inline int utf8_decode(const uint8_t* p, int64_t n, uint32_t* cp) {
  if (n <= 0) return 0;
  const uint8_t b = p[0];
  int len = b < 0x80 ? 1 : (b < 0xE0 ? 2 : (b < 0xF0 ? 3 : 4));
  if (len > n) len = (int)n;
  uint32_t c = len == 1 ? b : (b & (0xFF >> (len + 1)));
  for (int i = 1; i < len; ++i) c = (c << 6) | (p[i] & 0x3F);
  *cp = c;
  return len;
}

// regex_like (predicate.rs:246-306), kept as tokens instead of a regex string
LikeRegex regex_like(const std::string& pattern) {
  LikeRegex r;
  std::vector<uint32_t> chars;
  for (int64_t i = 0; i < (int64_t)pattern.size();) {
    uint32_t cp = 0;
    i += utf8_decode((const uint8_t*)pattern.data() + i, (int64_t)pattern.size() - i, &cp);
    chars.push_back(cp);
  }
  size_t i = 0;
  if (!chars.empty() && chars[0] == '%') {  // a leading % drops the `^` instead of emitting `^.*`
    r.anchored_start = false;
    i = 1;
  }
  for (; i < chars.size(); ++i) {
    const uint32_t c = chars[i];
    if (c == '\\') {
      if (i + 1 < chars.size()) r.toks.push_back({0, chars[++i]});
      else r.toks.push_back({0, '\\'});  // trailing backslash: literal
    } else if (c == '%') {
      r.toks.push_back({2, 0});
    } else if (c == '_') {
      r.toks.push_back({1, 0});
    } else {
      r.toks.push_back({0, c});
    }
  }
  if (!r.toks.empty() && r.toks.back().kind == 2) {  // a trailing `.*` is dropped together with the `$`
    r.toks.pop_back();
    r.anchored_end = false;
  }
  return r;
}

// `dead` memoises (token, position) pairs already shown not to match, which keeps the backtracking polynomial
// (the regex crate the reference uses is linear-time; an exponential oracle would only be a test hazard)
bool regex_match_here(const LikeRegex& r, size_t ti, const uint8_t* s, int64_t n, int64_t si, std::vector<uint8_t>& dead) {
  if (ti == r.toks.size()) return !r.anchored_end || si == n;
  uint8_t& d = dead[ti * (size_t)(n + 1) + (size_t)si];
  if (d) return false;
  const auto& t = r.toks[ti];
  bool ok = false;
  if (t.kind == 2) {
    for (int64_t k = si;;) {
      if (regex_match_here(r, ti + 1, s, n, k, dead)) {
        ok = true;
        break;
      }
      if (k >= n) break;
      uint32_t cp;
      k += utf8_decode(s + k, n - k, &cp);
    }
  } else if (si < n) {
    uint32_t cp;
    const int l = utf8_decode(s + si, n - si, &cp);
    ok = (t.kind == 1 || cp == t.cp) && regex_match_here(r, ti + 1, s, n, si + l, dead);
  }
  if (!ok) d = 1;
  return ok;
}

bool regex_is_match(const LikeRegex& r, const uint8_t* s, int64_t n) {
  std::vector<uint8_t> dead((r.toks.size() + 1) * (size_t)(n + 1), 0);
  if (r.anchored_start) return regex_match_here(r, 0, s, n, 0, dead);
  for (int64_t k = 0;;) {  // unanchored: a match may start at any character
    if (regex_match_here(r, 0, s, n, k, dead)) return true;
    if (k >= n) return false;
    uint32_t cp;
    k += utf8_decode(s + k, n - k, &cp);
  }
}

bool contains_like_pattern(const std::string& p) { return p.find_first_of("%_\\") != std::string::npos; }

struct LikePredicate {  // Predicate (predicate.rs:28-42)
  enum { EQ, STARTS, ENDS, CONTAINS, REGEX } kind = EQ;
  std::string needle;
  LikeRegex re;
  bool evaluate(const uint8_t* s, int64_t n) const {
    const int64_t m = (int64_t)needle.size();
    switch (kind) {
      case EQ: return n == m && (m == 0 || memcmp(s, needle.data(), (size_t)m) == 0);
      case STARTS: return m <= n && (m == 0 || memcmp(s, needle.data(), (size_t)m) == 0);
      case ENDS: return m <= n && (m == 0 || memcmp(s + n - m, needle.data(), (size_t)m) == 0);
      case CONTAINS:
        if (m == 0) return true;
        for (int64_t i = 0; i + m <= n; ++i)
          if (memcmp(s + i, needle.data(), (size_t)m) == 0) return true;
        return false;
      default: return regex_is_match(re, s, n);
    }
  }
};

// Predicate::like (predicate.rs:46-64)
LikePredicate predicate_like(const std::string& p) {
  LikePredicate r;
  auto ends_pct = [](const std::string& x) { return !x.empty() && x.back() == '%'; };
  auto starts_pct = [](const std::string& x) { return !x.empty() && x.front() == '%'; };
  if (!contains_like_pattern(p)) {
    r.kind = LikePredicate::EQ;
    r.needle = p;
  } else if (ends_pct(p) && !contains_like_pattern(p.substr(0, p.size() - 1))) {
    r.kind = LikePredicate::STARTS;
    r.needle = p.substr(0, p.size() - 1);
  } else if (starts_pct(p) && !contains_like_pattern(p.substr(1))) {
    r.kind = LikePredicate::ENDS;
    r.needle = p.substr(1);
  } else if (starts_pct(p) && ends_pct(p) && p.size() >= 2 && !contains_like_pattern(p.substr(1, p.size() - 2))) {
    r.kind = LikePredicate::CONTAINS;
    r.needle = p.substr(1, p.size() - 2);
  } else {
    r.kind = LikePredicate::REGEX;
    r.re = regex_like(p);
  }
  return r;
}
}  // namespace

// op: 0 like, 1 nlike, 2 starts_with, 3 ends_with, 4 contains; `pattern` is a length-1 array (Scalar)
int32_t orc_string_like(int32_t op, const orc_view* v, const orc_view* pattern, orc_out* out) {
  out_init(out);
  static const char* names[] = {"LIKE", "NLIKE", "STARTS_WITH", "ENDS_WITH", "CONTAINS"};
  if (!(v->type == ORC_UTF8 || v->type == ORC_LARGE_UTF8) || pattern->type != v->type)
    return fail(ORC_INVALID_ARGUMENT, "Invalid string/binary operation: %s %s %s", type_name(v->type), names[op],
                type_name(pattern->type));
  const int64_t len = v->length;
  out->type = ORC_BOOL;
  out->length = len;
  if (len == 0) return ORC_OK;
  const size_t bytes = bitmap_bytes(len);
  uint8_t* vals = (uint8_t*)xalloc(bytes);
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (pattern->validity && !get_bit(pattern->validity, pattern->validity_bit_offset)) {  // new_null(len) (like.rs:314)
    out->validity = (uint8_t*)xalloc(bytes);
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len;
    return ORC_OK;
  }
  const int ow = v->type == ORC_UTF8 ? 4 : 8;
  auto off = [ow](const orc_view* a, int64_t i) -> int64_t {
    return ow == 4 ? (int64_t)((const int32_t*)a->offsets)[i] : ((const int64_t*)a->offsets)[i];
  };
  const std::string pat((const char*)pattern->values + off(pattern, 0), (size_t)(off(pattern, 1) - off(pattern, 0)));
  LikePredicate pred;
  bool neg = false;
  switch (op) {  // op_scalar (like.rs:349-361)
    case 0: pred = predicate_like(pat); break;
    case 1: pred = predicate_like(pat); neg = true; break;
    case 2: pred.kind = LikePredicate::STARTS; pred.needle = pat; break;
    case 3: pred.kind = LikePredicate::ENDS; pred.needle = pat; break;
    default: pred.kind = LikePredicate::CONTAINS; pred.needle = pat; break;
  }
  for (int64_t i = 0; i < len; ++i) {  // BooleanArray::from_unary: every slot is evaluated, nulls are cloned
    const int64_t a = off(v, i), b = off(v, i + 1);
    if (pred.evaluate((const uint8_t*)v->values + a, b - a) != neg) set_bit(vals, i);
  }
  if (v->validity) {
    out->validity = nulls_clone(v, len);
    out->validity_bytes = (int64_t)bytes;
    out->null_count = len - count_set_bits(out->validity, 0, len);
  }
  return ORC_OK;
}

// length / bit_length (arrow-string/src/length.rs:35-46,:58-110)
int32_t orc_string_length(const orc_view* v, int32_t bits, orc_out* out) {
  out_init(out);
  if (!(v->type == ORC_UTF8 || v->type == ORC_LARGE_UTF8))
    return fail(ORC_COMPUTE_ERROR, "length not supported for %s", type_name(v->type));
  const int64_t len = v->length;
  const bool large = v->type == ORC_LARGE_UTF8;
  out->type = large ? ORC_INT64 : ORC_INT32;
  out->length = len;
  if (len == 0) return ORC_OK;
  const size_t vb = (size_t)len * (large ? 8 : 4);
  void* o = xalloc(vb);
  for (int64_t i = 0; i < len; ++i) {
    if (large) ((int64_t*)o)[i] = (((const int64_t*)v->offsets)[i + 1] - ((const int64_t*)v->offsets)[i]) * (bits ? 8 : 1);
    else ((int32_t*)o)[i] = (((const int32_t*)v->offsets)[i + 1] - ((const int32_t*)v->offsets)[i]) * (bits ? 8 : 1);
  }
  out->values = o;
  out->values_bytes = (int64_t)vb;
  if (v->validity) {
    out->validity = nulls_clone(v, len);
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(out->validity, 0, len);
  }
  return ORC_OK;
}

// nullif (arrow-select/src/nullif.rs:60-121); flags 2 = values/offsets borrowed, validity owned
int32_t orc_nullif(const orc_view* l, const orc_view* r, orc_out* out) {
  out_init(out);
  if (l->length != r->length)
    return fail(ORC_COMPUTE_ERROR, "Cannot perform comparison operation on arrays of different length");
  int64_t len = l->length;
  out->type = l->type;
  out->length = len;
  out->values = const_cast<void*>(l->values);
  out->values_bit_offset = l->values_bit_offset;
  out->offsets = const_cast<void*>(l->offsets);
  out->flags = 2;
  if (len == 0) {
    out->flags = 1;
    out->validity = const_cast<uint8_t*>(l->validity);
    out->validity_bit_offset = l->validity_bit_offset;
    return ORC_OK;
  }
  uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
  const uint8_t* rv = (const uint8_t*)r->values;
  for (int64_t i = 0; i < len; ++i) {
    bool lv = !l->validity || get_bit(l->validity, l->validity_bit_offset + i);
    bool rr = get_bit(rv, r->values_bit_offset + i) && (!r->validity || get_bit(r->validity, r->validity_bit_offset + i));
    if (lv && !rr) set_bit(nb, i);
  }
  out->validity = nb;
  out->validity_bytes = (int64_t)bitmap_bytes(len);
  out->null_count = len - count_set_bits(nb, 0, len);
  return ORC_OK;
}

int32_t orc_cast(const orc_view* in, int32_t to, int32_t safe, orc_out* out) {
  out_init(out);
  if ((in->type == ORC_UTF8 || in->type == ORC_LARGE_UTF8) && to != ORC_UTF8 && to != ORC_LARGE_UTF8) {
    switch (to) {
      case ORC_INT8: return parse_string<int8_t>(in, to, safe, out);
      case ORC_INT16: return parse_string<int16_t>(in, to, safe, out);
      case ORC_INT32: return parse_string<int32_t>(in, to, safe, out);
      case ORC_INT64: return parse_string<int64_t>(in, to, safe, out);
      case ORC_UINT8: return parse_string<uint8_t>(in, to, safe, out);
      case ORC_UINT16: return parse_string<uint16_t>(in, to, safe, out);
      case ORC_UINT32: return parse_string<uint32_t>(in, to, safe, out);
      case ORC_UINT64: return parse_string<uint64_t>(in, to, safe, out);
      case ORC_FLOAT32: return parse_string<float>(in, to, safe, out);
      case ORC_FLOAT64: return parse_string<double>(in, to, safe, out);
    }
    return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
  }
  if (in->type == ORC_BOOL && to != ORC_BOOL) {
    switch (to) {
      case ORC_INT8: return cast_bool_to_num<int8_t>(in, to, out);
      case ORC_INT16: return cast_bool_to_num<int16_t>(in, to, out);
      case ORC_INT32: return cast_bool_to_num<int32_t>(in, to, out);
      case ORC_INT64: return cast_bool_to_num<int64_t>(in, to, out);
      case ORC_UINT8: return cast_bool_to_num<uint8_t>(in, to, out);
      case ORC_UINT16: return cast_bool_to_num<uint16_t>(in, to, out);
      case ORC_UINT32: return cast_bool_to_num<uint32_t>(in, to, out);
      case ORC_UINT64: return cast_bool_to_num<uint64_t>(in, to, out);
      case ORC_FLOAT32: return cast_bool_to_num<float>(in, to, out);
      case ORC_FLOAT64: return cast_bool_to_num<double>(in, to, out);
    }
    return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
  }
  if (to == ORC_BOOL && in->type != ORC_BOOL) {
    switch (in->type) {
      case ORC_INT8: return cast_num_to_bool<int8_t>(in, out);
      case ORC_INT16: return cast_num_to_bool<int16_t>(in, out);
      case ORC_INT32: return cast_num_to_bool<int32_t>(in, out);
      case ORC_INT64: return cast_num_to_bool<int64_t>(in, out);
      case ORC_UINT8: return cast_num_to_bool<uint8_t>(in, out);
      case ORC_UINT16: return cast_num_to_bool<uint16_t>(in, out);
      case ORC_UINT32: return cast_num_to_bool<uint32_t>(in, out);
      case ORC_UINT64: return cast_num_to_bool<uint64_t>(in, out);
      case ORC_FLOAT32: return cast_num_to_bool<float>(in, out);
      case ORC_FLOAT64: return cast_num_to_bool<double>(in, out);
    }
    return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
  }
  if (to == ORC_UTF8) return cast_to_string_dispatch<int32_t>(in, to, out);
  if (to == ORC_LARGE_UTF8) return cast_to_string_dispatch<int64_t>(in, to, out);
  if (in->type == to) {  // cast_with_options :797-799 — clone
    int w = type_width(to);
    if (w <= 0) return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
    out->type = to;
    out->length = in->length;
    out->values = xalloc((size_t)in->length * w);
    memcpy(out->values, in->values, (size_t)in->length * w);
    out->values_bytes = in->length * w;
    if (in->validity) {
      out->validity = nulls_clone(in, in->length);
      out->validity_bytes = (int64_t)bitmap_bytes(in->length);
      out->null_count = resolve_nulls(in);
    }
    return ORC_OK;
  }
  switch (in->type) {
    case ORC_INT8: return cast_from<int8_t>(in, to, safe, out);
    case ORC_INT16: return cast_from<int16_t>(in, to, safe, out);
    case ORC_INT32: return cast_from<int32_t>(in, to, safe, out);
    case ORC_INT64: return cast_from<int64_t>(in, to, safe, out);
    case ORC_UINT8: return cast_from<uint8_t>(in, to, safe, out);
    case ORC_UINT16: return cast_from<uint16_t>(in, to, safe, out);
    case ORC_UINT32: return cast_from<uint32_t>(in, to, safe, out);
    case ORC_UINT64: return cast_from<uint64_t>(in, to, safe, out);
    case ORC_FLOAT16: return cast_from<F16>(in, to, safe, out);
    case ORC_FLOAT32: return cast_from<float>(in, to, safe, out);
    case ORC_FLOAT64: return cast_from<double>(in, to, safe, out);
  }
  return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", type_name(in->type), type_name(to));
}

}  // extern "C"

// ------------------------------------------------------------ temporal casts
// cast_with_options, the "temporal casts" arms (arrow-cast/src/cast/mod.rs:1700-2260), restated arm by arm with
// the reference's own recursion (`cast_with_options(&cast_with_options(array, &Int32)?, &Int64)`), on top of literal
// restatements of PrimitiveArray::unary / unary_opt / try_unary (arrow-array/src/array/primitive_array.rs:861,
// :1065, :990).  Calendar: chrono 0.4.45 (Cargo.lock:854; not under /root/reference) — DateTime::from_timestamp
// splits seconds with div_euclid / rem_euclid and accepts the day iff its proleptic-Gregorian YEAR lies in
// NaiveDate's MIN_YEAR..=MAX_YEAR = -262143..=262142; here the year is recovered with the civil-from-days
// algorithm (Howard Hinnant's public-domain date algorithms), not with precomputed day bounds.
namespace {

enum { DT_DATE32 = 32, DT_DATE64 = 33, DT_TIME32 = 34, DT_TIME64 = 35, DT_TIMESTAMP = 36, DT_DURATION = 37 };
enum { U_S = 0, U_MS = 1, U_US = 2, U_NS = 3 };
const int64_t MILLISECONDS = 1000, MICROSECONDS = 1000000, NANOSECONDS = 1000000000;
const int64_t SECONDS_IN_DAY = 86400, MILLISECONDS_IN_DAY = SECONDS_IN_DAY * MILLISECONDS,
              MICROSECONDS_IN_DAY = SECONDS_IN_DAY * MICROSECONDS, NANOSECONDS_IN_DAY = SECONDS_IN_DAY * NANOSECONDS;

int64_t div_euclid(int64_t a, int64_t b) { int64_t q = a / b; return (a % b < 0) ? q - 1 : q; }  // b > 0
int64_t rem_euclid(int64_t a, int64_t b) { int64_t r = a % b; return r < 0 ? r + b : r; }

int64_t year_of_day(int64_t days_since_epoch) {  // civil_from_days, year only
  __int128 z = (__int128)days_since_epoch + 719468;
  __int128 era = (z >= 0 ? z : z - 146096) / 146097;
  int64_t doe = (int64_t)(z - era * 146097);
  int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  int64_t mp = (5 * doy + 2) / 153;
  int64_t m = mp < 10 ? mp + 3 : mp - 9;
  return (int64_t)(yoe + era * 400) + (m <= 2);
}

struct NaiveDateTime { int64_t day; int64_t sod; int64_t nanos; };  // days since 1970-01-01, second of day, nanosecond

// chrono DateTime::from_timestamp(secs, nsecs)
bool from_timestamp(int64_t secs, int64_t nsecs, NaiveDateTime* o) {
  int64_t days = div_euclid(secs, 86400);
  int64_t y = year_of_day(days);
  if (y < -262143 || y > 262142) return false;
  *o = {days, rem_euclid(secs, 86400), nsecs};
  return true;
}
// as_datetime::<TimestampXType> (temporal_conversions.rs:141-216)
bool as_datetime_ts(int unit, int64_t v, NaiveDateTime* o) {
  switch (unit) {
    case U_S: return from_timestamp(v, 0, o);
    case U_MS: return from_timestamp(div_euclid(v, MILLISECONDS), rem_euclid(v, MILLISECONDS) * MICROSECONDS, o);
    case U_US: return from_timestamp(div_euclid(v, MICROSECONDS), rem_euclid(v, MICROSECONDS) * MILLISECONDS, o);
    default: return from_timestamp(div_euclid(v, NANOSECONDS), rem_euclid(v, NANOSECONDS), o);
  }
}
// Utc.from_utc_datetime(&naive).with_timezone(&tz) for a fixed offset: the local wall clock
NaiveDateTime with_offset(NaiveDateTime d, int64_t off) {
  int64_t s = d.day * 86400 + d.sod + off;
  return {div_euclid(s, 86400), rem_euclid(s, 86400), d.nanos};
}

const char* ts_type_name(int unit) {
  static const char* n[] = {"arrow_array::types::TimestampSecondType", "arrow_array::types::TimestampMillisecondType",
                            "arrow_array::types::TimestampMicrosecondType", "arrow_array::types::TimestampNanosecondType"};
  return n[unit];
}
const char* unit_disp(int u) { static const char* n[] = {"s", "ms", "\xC2\xB5s", "ns"}; return (u >= 0 && u < 4) ? n[u] : "?"; }
std::string dt_text(const orc_data_type* t) {
  char b[96];
  switch (t->id) {
    case DT_DATE32: return "Date32";
    case DT_DATE64: return "Date64";
    case DT_TIME32: snprintf(b, sizeof b, "Time32(%s)", unit_disp(t->unit)); return b;
    case DT_TIME64: snprintf(b, sizeof b, "Time64(%s)", unit_disp(t->unit)); return b;
    case DT_DURATION: snprintf(b, sizeof b, "Duration(%s)", unit_disp(t->unit)); return b;
    case DT_TIMESTAMP:
      if (t->has_tz) {
        int o = t->tz_offset_seconds, ao = o < 0 ? -o : o;
        snprintf(b, sizeof b, "Timestamp(%s, \"%c%02d:%02d\")", unit_disp(t->unit), o < 0 ? '-' : '+', ao / 3600, ao / 60 % 60);
      } else snprintf(b, sizeof b, "Timestamp(%s)", unit_disp(t->unit));
      return b;
    default: return type_name(t->id);
  }
}

template <typename O>
O* start_out(const orc_view* in, int32_t to, orc_out* out) {
  O* ov = (O*)xalloc((size_t)in->length * sizeof(O));
  out->type = to;
  out->length = in->length;
  out->values = ov;
  out->values_bytes = in->length * (int64_t)sizeof(O);
  return ov;
}
void attach_nulls(orc_out* out, uint8_t* nb, int64_t len) {
  if (!nb) return;
  out->validity = nb;
  out->validity_bytes = (int64_t)bitmap_bytes(len);
  out->null_count = len - count_set_bits(nb, 0, len);
}

// PrimitiveArray::unary (:861): `op` on EVERY slot, nulls cloned
template <typename I, typename O, typename F>
int32_t prim_unary(const orc_view* in, int32_t to, orc_out* out, F op) {
  const I* iv = (const I*)in->values;
  O* ov = start_out<O>(in, to, out);
  for (int64_t i = 0; i < in->length; ++i) ov[i] = op(iv[i]);
  attach_nulls(out, nulls_clone(in, in->length), in->length);
  return ORC_OK;
}
// PrimitiveArray::unary_opt (:1065): valid slots only, None becomes null, a null buffer is always attached
template <typename I, typename O, typename F>
int32_t prim_unary_opt(const orc_view* in, int32_t to, orc_out* out, F op) {
  const int64_t len = in->length;
  const I* iv = (const I*)in->values;
  O* ov = start_out<O>(in, to, out);
  uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
  for (int64_t i = 0; i < len; ++i) {
    if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
    O o;
    if (op(iv[i], &o)) {
      ov[i] = o;
      set_bit(nb, i);
    }
  }
  out->validity = nb;
  out->validity_bytes = (int64_t)bitmap_bytes(len);
  out->null_count = len - count_set_bits(nb, 0, len);
  return ORC_OK;
}
// PrimitiveArray::try_unary (:990): valid slots only, first Err returned, nulls cloned
template <typename I, typename O, typename F>
int32_t prim_try_unary(const orc_view* in, int32_t to, orc_out* out, F op) {
  const int64_t len = in->length;
  const I* iv = (const I*)in->values;
  O* ov = start_out<O>(in, to, out);
  for (int64_t i = 0; i < len; ++i) {
    if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
    int32_t st = op(iv[i], &ov[i]);
    if (st != ORC_OK) {
      orc_release(out);
      return st;
    }
  }
  attach_nulls(out, nulls_clone(in, len), len);
  return ORC_OK;
}

orc_view view_of_out(const orc_out* o) {
  orc_view v{};
  v.type = o->type;
  v.length = o->length;
  v.null_count = o->validity ? o->null_count : 0;
  v.values = o->values;
  v.validity = o->validity;
  v.validity_bit_offset = o->validity_bit_offset;
  return v;
}

// checked_mul for safe, mul_checked (ArithmeticOverflow text) otherwise, in the width of O
template <typename I, typename O>
int32_t scale_up(const orc_view* in, int32_t to, int64_t k, bool safe, orc_out* out) {
  auto mul = [k](I x, O* o) {
    __int128 p = (__int128)x * (__int128)k;
    if (p < (__int128)std::numeric_limits<O>::min() || p > (__int128)std::numeric_limits<O>::max()) return false;
    *o = (O)p;
    return true;
  };
  if (safe) return prim_unary_opt<I, O>(in, to, out, mul);
  return prim_try_unary<I, O>(in, to, out, [&](I x, O* o) -> int32_t {
    if (mul(x, o)) return ORC_OK;
    return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %lld * %lld", (long long)x, (long long)k);
  });
}

bool needs_unit(int id) { return id == DT_TIME32 || id == DT_TIME64 || id == DT_TIMESTAMP || id == DT_DURATION; }
bool unit_valid(const orc_data_type* t) {
  if (t->unit < U_S || t->unit > U_NS) return false;
  if (t->id == DT_TIME32) return t->unit <= U_MS;
  if (t->id == DT_TIME64) return t->unit >= U_US;
  return true;
}
bool dt_is_numeric(int id) { return is_integer(id) || id == ORC_FLOAT32 || id == ORC_FLOAT64; }
int64_t time_unit_multiple(int u) { static const int64_t m[] = {1, 1000, 1000000, 1000000000}; return m[u]; }
orc_data_type plain(int id) { orc_data_type t{}; t.id = id; return t; }
orc_data_type with_unit(int id, int unit) { orc_data_type t{}; t.id = id; t.unit = unit; return t; }
bool dt_equal(const orc_data_type* a, const orc_data_type* b) {
  if (a->id != b->id) return false;
  if (!needs_unit(a->id)) return true;
  if (a->unit != b->unit) return false;
  if (a->id == DT_TIMESTAMP) return a->has_tz == b->has_tz && (!a->has_tz || a->tz_offset_seconds == b->tz_offset_seconds);
  return true;
}

int32_t cast_temporal(const orc_view* in, const orc_data_type* from, const orc_data_type* to, bool safe, orc_out* out);

// `cast_with_options(&cast_with_options(array, &mid, ..)?, to, ..)`
int32_t cast_via(const orc_view* in, const orc_data_type* from, orc_data_type mid, const orc_data_type* to, bool safe,
                 orc_out* out) {
  orc_out tmp;
  int32_t st = cast_temporal(in, from, &mid, safe, &tmp);
  if (st != ORC_OK) return st;
  orc_view v = view_of_out(&tmp);
  st = cast_temporal(&v, &mid, to, safe, out);
  orc_release(&tmp);
  return st;
}
// a computed array (already in `tmp`, of logical type `mid`) handed to `cast_with_options(&array, to_type, ..)`
int32_t then_cast(orc_out* tmp, orc_data_type mid, const orc_data_type* to, bool safe, orc_out* out) {
  orc_view v = view_of_out(tmp);
  int32_t st = cast_temporal(&v, &mid, to, safe, out);
  orc_release(tmp);
  return st;
}

// adjust_timestamp_to_timezone (mod.rs:2629-2649) for a fixed-offset zone
int32_t adjust_timestamp_to_timezone(const orc_view* in, int unit, int64_t off, bool safe, orc_out* out) {
  auto adjust = [unit, off](int64_t o, int64_t* r) {
    NaiveDateTime local;
    if (!as_datetime_ts(unit, o, &local)) return false;
    // local - offset.fix(): must stay a NaiveDateTime (the reference would panic past the calendar's ends)
    NaiveDateTime shifted = with_offset(local, -off);
    int64_t y = year_of_day(shifted.day);
    if (y < -262143 || y > 262142) return false;
    // T::from_naive_datetime(.., None): checked arithmetic back into the unit
    __int128 v = (__int128)o - (__int128)off * time_unit_multiple(unit);
    if (v < (__int128)INT64_MIN || v > (__int128)INT64_MAX) return false;
    *r = (int64_t)v;
    return true;
  };
  if (safe) return prim_unary_opt<int64_t, int64_t>(in, ORC_INT64, out, adjust);
  return prim_try_unary<int64_t, int64_t>(in, ORC_INT64, out, [&](int64_t o, int64_t* r) -> int32_t {
    if (adjust(o, r)) return ORC_OK;
    return fail(ORC_CAST_ERROR, "Cannot cast timezone to different timezone");
  });
}

int32_t cast_temporal(const orc_view* in, const orc_data_type* from, const orc_data_type* to, bool safe, orc_out* out) {
  out_init(out);
  const int F = from->id, T = to->id;
  auto unsupported = [&] {
    return fail(ORC_CAST_ERROR, "Casting from %s to %s not supported", dt_text(from).c_str(), dt_text(to).c_str());
  };
  if ((needs_unit(F) && !unit_valid(from)) || (needs_unit(T) && !unit_valid(to))) return unsupported();
  const bool f_temporal = F >= DT_DATE32, t_temporal = T >= DT_DATE32;
  if (!f_temporal && !t_temporal) return orc_cast(in, T, safe, out);
  auto reinterpret = [&](int32_t phys) { return orc_cast(in, phys, safe, out); };  // same-width clone
  if (dt_equal(from, to)) return reinterpret(in->type);  // mod.rs:797-799

  // mod.rs:1701-1761
  if (F == ORC_INT32 && (T == DT_DATE32 || T == DT_TIME32)) return reinterpret(ORC_INT32);
  if (F == ORC_INT32 && T == DT_DATE64) return cast_via(in, from, plain(DT_DATE32), to, safe, out);
  if ((F == DT_DATE32 || F == DT_TIME32) && T == ORC_INT32) return reinterpret(ORC_INT32);
  if ((F == DT_DATE32 || F == DT_TIME32) && T == ORC_INT64) return cast_via(in, from, plain(ORC_INT32), to, safe, out);
  if (F == ORC_INT64 && (T == DT_DATE64 || T == DT_TIME64)) return reinterpret(ORC_INT64);
  if (F == ORC_INT64 && T == DT_DATE32) return cast_via(in, from, plain(ORC_INT32), to, safe, out);
  if ((F == DT_DATE64 || F == DT_TIME64) && T == ORC_INT64) return reinterpret(ORC_INT64);
  if (F == DT_DATE64 && T == ORC_INT32) return cast_via(in, from, plain(ORC_INT64), to, safe, out);
  // mod.rs:1762-1781
  if (F == DT_DATE32 && T == DT_DATE64)
    return prim_unary<int32_t, int64_t>(in, ORC_INT64, out, [](int32_t x) { return (int64_t)((uint64_t)(int64_t)x * (uint64_t)MILLISECONDS_IN_DAY); });
  if (F == DT_DATE64 && T == DT_DATE32) {
    auto f = [](int64_t x, int32_t* o) {
      int64_t q = x / MILLISECONDS_IN_DAY;
      if (q < INT32_MIN || q > INT32_MAX) return false;
      *o = (int32_t)q;
      return true;
    };
    if (safe) return prim_unary_opt<int64_t, int32_t>(in, ORC_INT32, out, f);
    return prim_try_unary<int64_t, int32_t>(in, ORC_INT32, out, [&](int64_t x, int32_t* o) -> int32_t {
      if (f(x, o)) return ORC_OK;
      return fail(ORC_CAST_ERROR, "Cannot cast Date64 value %lld to Date32 without overflow", (long long)x);
    });
  }
  // mod.rs:1783-1851, one arm per unit pair
  if (F == DT_TIME32 && T == DT_TIME32) {
    if (from->unit == U_S) return scale_up<int32_t, int32_t>(in, ORC_INT32, MILLISECONDS, safe, out);
    return prim_unary<int32_t, int32_t>(in, ORC_INT32, out, [](int32_t x) { return x / (int32_t)MILLISECONDS; });
  }
  if (F == DT_TIME32 && T == DT_TIME64) {
    int64_t k = from->unit == U_S ? (to->unit == U_US ? MICROSECONDS : NANOSECONDS)
                                  : (to->unit == U_US ? MICROSECONDS / MILLISECONDS : NANOSECONDS / MILLISECONDS);
    return prim_unary<int32_t, int64_t>(in, ORC_INT64, out, [k](int32_t x) { return (int64_t)x * k; });
  }
  if (F == DT_TIME64 && T == DT_TIME32) {
    int64_t k = from->unit == U_US ? (to->unit == U_S ? MICROSECONDS : MICROSECONDS / MILLISECONDS)
                                   : (to->unit == U_S ? NANOSECONDS : NANOSECONDS / MILLISECONDS);
    return prim_unary<int64_t, int32_t>(in, ORC_INT32, out, [k](int64_t x) { return (int32_t)(uint32_t)(uint64_t)(x / k); });
  }
  if (F == DT_TIME64 && T == DT_TIME64) {
    if (from->unit == U_US)  // `x * (NANOSECONDS / MICROSECONDS)`: a release build wraps
      return prim_unary<int64_t, int64_t>(in, ORC_INT64, out, [](int64_t x) { return (int64_t)((uint64_t)x * 1000u); });
    return prim_unary<int64_t, int64_t>(in, ORC_INT64, out, [](int64_t x) { return x / (NANOSECONDS / MICROSECONDS); });
  }
  // mod.rs:1854-1878: Timestamp -> numeric reinterprets as Int64 first; numeric -> Timestamp casts to Int64 first
  if ((F == DT_TIMESTAMP || F == DT_DURATION) && dt_is_numeric(T)) return orc_cast(in, T, safe, out);
  if (dt_is_numeric(F) && (T == DT_TIMESTAMP || T == DT_DURATION)) return orc_cast(in, ORC_INT64, safe, out);
  // mod.rs:1880-1937 / :2257-2279
  if ((F == DT_TIMESTAMP && T == DT_TIMESTAMP) || (F == DT_DURATION && T == DT_DURATION)) {
    const int64_t from_size = time_unit_multiple(from->unit), to_size = time_unit_multiple(to->unit);
    orc_out conv;
    out_init(&conv);
    int32_t st;
    if (from_size > to_size) {
      const int64_t divisor = from_size / to_size;
      st = prim_unary<int64_t, int64_t>(in, ORC_INT64, &conv, [divisor](int64_t o) { return o / divisor; });
    } else if (from_size == to_size) {
      st = orc_cast(in, ORC_INT64, safe, &conv);
    } else {
      st = scale_up<int64_t, int64_t>(in, ORC_INT64, to_size / from_size, safe, &conv);
    }
    if (st != ORC_OK) return st;
    if (F == DT_TIMESTAMP && !from->has_tz && to->has_tz) {
      orc_view v = view_of_out(&conv);
      st = adjust_timestamp_to_timezone(&v, to->unit, to->tz_offset_seconds, safe, out);
      orc_release(&conv);
      return st;
    }
    *out = conv;
    return ORC_OK;
  }
  if (F == DT_TIMESTAMP && T == DT_DATE32) {  // timestamp_to_date32, mod.rs:633-659
    const int unit = from->unit;
    const int64_t off = from->has_tz ? from->tz_offset_seconds : 0;
    return prim_try_unary<int64_t, int32_t>(in, ORC_INT32, out, [&](int64_t x, int32_t* o) -> int32_t {
      NaiveDateTime d;
      if (!as_datetime_ts(unit, x, &d)) return fail(ORC_CAST_ERROR, "Cannot convert %s %lld to datetime", ts_type_name(unit), (long long)x);
      *o = (int32_t)with_offset(d, off).day;  // Date32Type::from_naive_date
      return ORC_OK;
    });
  }
  if (F == DT_TIMESTAMP && T == DT_DATE64) {  // mod.rs:1950-1973
    switch (from->unit) {
      case U_S: return scale_up<int64_t, int64_t>(in, ORC_INT64, MILLISECONDS, safe, out);
      case U_MS: return reinterpret(ORC_INT64);
      case U_US: return prim_unary<int64_t, int64_t>(in, ORC_INT64, out, [](int64_t x) { return x / (MICROSECONDS / MILLISECONDS); });
      default: return prim_unary<int64_t, int64_t>(in, ORC_INT64, out, [](int64_t x) { return x / (NANOSECONDS / MILLISECONDS); });
    }
  }
  if (F == DT_TIMESTAMP && (T == DT_TIME32 || T == DT_TIME64)) {  // mod.rs:1974-2165
    const int unit = from->unit, tunit = to->unit;
    const int64_t off = from->has_tz ? from->tz_offset_seconds : 0;
    auto time_of = [&](int64_t x, NaiveDateTime* t) -> int32_t {  // as_time_res_with_timezone :615-631
      NaiveDateTime d;
      if (!as_datetime_ts(unit, x, &d)) return fail(ORC_CAST_ERROR, "Failed to create naive time with %s %lld", ts_type_name(unit), (long long)x);
      *t = with_offset(d, off);
      return ORC_OK;
    };
    if (T == DT_TIME32)
      return prim_try_unary<int64_t, int32_t>(in, ORC_INT32, out, [&](int64_t x, int32_t* o) -> int32_t {
        NaiveDateTime t{};
        int32_t st = time_of(x, &t);
        if (st != ORC_OK) return st;
        // time_to_time32s / time_to_time32ms (temporal_conversions.rs:113-124)
        *o = tunit == U_S ? (int32_t)t.sod : (int32_t)(t.sod * MILLISECONDS + t.nanos * MILLISECONDS / NANOSECONDS);
        return ORC_OK;
      });
    return prim_try_unary<int64_t, int64_t>(in, ORC_INT64, out, [&](int64_t x, int64_t* o) -> int32_t {
      NaiveDateTime t{};
      int32_t st = time_of(x, &t);
      if (st != ORC_OK) return st;
      // time_to_time64us / time_to_time64ns (:127-139)
      *o = tunit == U_US ? t.sod * MICROSECONDS + t.nanos * MICROSECONDS / NANOSECONDS : t.sod * NANOSECONDS + t.nanos;
      return ORC_OK;
    });
  }
  if (F == DT_DATE64 && T == DT_TIMESTAMP) {  // mod.rs:2166-2194
    orc_out tmp;
    out_init(&tmp);
    int32_t st;
    switch (to->unit) {
      case U_S: st = prim_unary<int64_t, int64_t>(in, ORC_INT64, &tmp, [](int64_t x) { return x / MILLISECONDS; }); break;
      case U_MS: st = orc_cast(in, ORC_INT64, safe, &tmp); break;
      case U_US: st = prim_unary<int64_t, int64_t>(in, ORC_INT64, &tmp, [](int64_t x) { return (int64_t)((uint64_t)x * (uint64_t)(MICROSECONDS / MILLISECONDS)); }); break;
      default: st = prim_unary<int64_t, int64_t>(in, ORC_INT64, &tmp, [](int64_t x) { return (int64_t)((uint64_t)x * (uint64_t)(NANOSECONDS / MILLISECONDS)); }); break;
    }
    if (st != ORC_OK) return st;
    return then_cast(&tmp, with_unit(DT_TIMESTAMP, to->unit), to, safe, out);
  }
  if (F == DT_DATE32 && T == DT_TIMESTAMP) {  // mod.rs:2195-2234
    orc_out tmp;
    out_init(&tmp);
    int32_t st;
    switch (to->unit) {
      case U_S: st = prim_unary<int32_t, int64_t>(in, ORC_INT64, &tmp, [](int32_t x) { return (int64_t)x * SECONDS_IN_DAY; }); break;
      case U_MS: st = prim_unary<int32_t, int64_t>(in, ORC_INT64, &tmp, [](int32_t x) { return (int64_t)x * MILLISECONDS_IN_DAY; }); break;
      case U_US: st = scale_up<int32_t, int64_t>(in, ORC_INT64, MICROSECONDS_IN_DAY, safe, &tmp); break;
      default: st = scale_up<int32_t, int64_t>(in, ORC_INT64, NANOSECONDS_IN_DAY, safe, &tmp); break;
    }
    if (st != ORC_OK) return st;
    return then_cast(&tmp, with_unit(DT_TIMESTAMP, to->unit), to, safe, out);
  }
  return unsupported();
}

}  // namespace

// --------------------------------------------------- temporal arithmetic (types)
// arithmetic_op's type rules for temporal operands (arrow-arith/src/numeric.rs:225-275) with timestamp_op
// (:426-537), duration_op (:877-895) and the Date - Date arms of date_op (:898-932).  Interval arms: not restated.
namespace {
enum { DT_INTERVAL = 38 };
const char* op_display(int op) {  // numeric.rs:203-213
  switch (op) { case 0: case 1: return "+"; case 2: case 3: return "-"; case 4: case 5: return "*"; case 6: return "/"; default: return "%"; }
}
std::string arith_dt_text(const orc_data_type* t) {
  if (t->id == 39) {
    char b[48];
    snprintf(b, sizeof b, "Decimal128(%d, %d)", t->precision, t->scale);
    return b;
  }
  if (t->id == DT_INTERVAL) {
    static const char* n[] = {"YearMonth", "DayTime", "MonthDayNano"};
    return std::string("Interval(") + n[t->unit % 3] + ")";
  }
  return dt_text(t);
}
// decimal_op (arrow-arith/src/numeric.rs:971-1103) on Decimal128, with the compiler's native __int128 arithmetic
// (__builtin_*_overflow = i128::checked_*): restated arm by arm, independent of the device's limb arithmetic.
typedef __int128 oi128;
std::string i128_dbg(oi128 v) {
  if (v == 0) return "0";
  unsigned __int128 u = v < 0 ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
  std::string digits;
  while (u) { digits.insert(digits.begin(), (char)('0' + (int)(u % 10))); u /= 10; }
  return (v < 0 ? "-" : "") + digits;
}
bool pow10_checked_orc(int64_t exp, oi128* out) {  // 10.pow_checked(exp as u32)
  oi128 v = 1;
  if (exp < 0) return false;  // a negative difference as u32 is astronomically large
  for (int64_t i = 0; i < exp; ++i) if (__builtin_mul_overflow(v, (oi128)10, &v)) return false;
  *out = v;
  return true;
}
oi128 pow10_wrapping_orc(uint32_t exp) {
  unsigned __int128 v = 1;
  for (uint32_t i = 0; i < exp && i < 200; ++i) v *= 10;
  return (oi128)v;
}
int sat8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
enum { DT_DECIMAL128 = 39 };

int32_t decimal_op(int op, const orc_view* l, bool l_s, const orc_data_type* lt, const orc_view* r, bool r_s,
                   const orc_data_type* rt, orc_out* out, orc_data_type* ot) {
  const int p1 = lt->precision, s1 = lt->scale, p2 = rt->precision, s2 = rt->scale;
  char lname[48], rname[48];
  snprintf(lname, sizeof lname, "Decimal128(%d, %d)", p1, s1);
  snprintf(rname, sizeof rname, "Decimal128(%d, %d)", p2, s2);
  const bool add = op == 0 || op == 1, sub = op == 2 || op == 3, mul = op == 4 || op == 5;
  oi128 l_mul = 1, r_mul = 1;
  int result_precision, result_scale;
  bool equal_scale_fast_path = false;
  if (add || sub) {
    result_scale = std::max(s1, s2);
    int whole = std::max((int)(int8_t)p1 - s1, (int)(int8_t)p2 - s2);
    result_precision = std::min(std::min((int)(uint8_t)(int8_t)sat8(result_scale + (int8_t)whole) + 1, 255), 38);
    if (!pow10_checked_orc(result_scale - s1, &l_mul)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: 10 ^ %d", result_scale - s1);
    if (!pow10_checked_orc(result_scale - s2, &r_mul)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: 10 ^ %d", result_scale - s2);
    equal_scale_fast_path = s1 == s2;
  } else if (mul) {
    result_precision = std::min(std::min(p1 + std::min(p2 + 1, 255), 255), 38);
    result_scale = sat8(s1 + s2);
    if (result_scale > 38)
      return fail(ORC_INVALID_ARGUMENT, "Output scale of %s %s %s would exceed max scale of 38", lname, op_display(op), rname);
  } else if (op == 6) {
    result_scale = std::min(sat8(s1 + 4), 38);
    int mul_pow = (int8_t)(result_scale - s1 + s2);
    result_precision = std::min((int)(uint8_t)(int8_t)sat8(mul_pow + (int8_t)p1), 38);
    if (mul_pow > 0) {
      if (!pow10_checked_orc(mul_pow, &l_mul)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: 10 ^ %d", mul_pow);
    } else if (mul_pow < 0) {
      int e = (uint8_t)(int8_t)(-mul_pow);
      if (!pow10_checked_orc(e, &r_mul)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: 10 ^ %d", e);
    }
  } else {
    result_scale = std::max(s1, s2);
    int whole = std::min((int)(int8_t)p1 - s1, (int)(int8_t)p2 - s2);
    result_precision = std::min((int)(uint8_t)(int8_t)sat8(result_scale + (int8_t)whole), 38);
    l_mul = pow10_wrapping_orc((uint32_t)(result_scale - s1));
    r_mul = pow10_wrapping_orc((uint32_t)(result_scale - s2));
  }
  // one row: l.mul_checked(l_mul)?.<op>_checked(r.mul_checked(r_mul)?)
  auto row = [&](oi128 a, oi128 b, oi128* o) -> int32_t {
    oi128 x = a, y = b;
    if (!mul && !equal_scale_fast_path) {
      if (__builtin_mul_overflow(a, l_mul, &x)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %s * %s", i128_dbg(a).c_str(), i128_dbg(l_mul).c_str());
      if (__builtin_mul_overflow(b, r_mul, &y)) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %s * %s", i128_dbg(b).c_str(), i128_dbg(r_mul).c_str());
    }
    bool ovf;
    if (add) ovf = __builtin_add_overflow(x, y, o);
    else if (sub) ovf = __builtin_sub_overflow(x, y, o);
    else if (mul) ovf = __builtin_mul_overflow(x, y, o);
    else {
      if (y == 0) return fail(ORC_DIVIDE_BY_ZERO, "Divide by zero error");
      const oi128 mn = (oi128)((unsigned __int128)1 << 127);
      ovf = (x == mn && y == -1);
      if (!ovf) *o = op == 6 ? x / y : x % y;
    }
    if (ovf) return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %s %s %s", i128_dbg(x).c_str(), op_display(op), i128_dbg(y).c_str());
    return ORC_OK;
  };
  const oi128* lv = (const oi128*)l->values;
  const oi128* rv = (const oi128*)r->values;
  auto load = [](const oi128* p, int64_t i) { oi128 v; memcpy(&v, (const char*)p + i * 16, 16); return v; };  // unaligned-safe
  int64_t len;
  uint8_t* nb = nullptr;
  if (l_s != r_s) {  // try_op! with one scalar side
    const orc_view* arr = l_s ? r : l;
    const orc_view* sc = l_s ? l : r;
    len = arr->length;
    out->type = ORC_FIXED16;
    out->length = len;
    out->values = xalloc((size_t)len * 16);
    out->values_bytes = len * 16;
    if (resolve_nulls(sc) != 0) {
      out->validity = (uint8_t*)xalloc(bitmap_bytes(len));
      out->validity_bytes = (int64_t)bitmap_bytes(len);
      out->null_count = len;
    } else {
      nb = nulls_clone(arr, len);
      for (int64_t i = 0; i < len; ++i) {
        if (nb && !get_bit(nb, i)) continue;
        oi128 o;
        int32_t st = l_s ? row(load(lv, 0), load(rv, i), &o) : row(load(lv, i), load(rv, 0), &o);
        if (st != ORC_OK) { free(nb); orc_release(out); return st; }
        memcpy((char*)out->values + i * 16, &o, 16);
      }
      attach_nulls(out, nb, len);
    }
  } else {
    if (l->length != r->length) return fail(ORC_COMPUTE_ERROR, "Cannot perform a binary operation on arrays of different length");
    len = l->length;
    out->type = ORC_FIXED16;
    out->length = len;
    out->values = xalloc((size_t)len * 16);
    out->values_bytes = len * 16;
    if (resolve_nulls(l) != 0 || resolve_nulls(r) != 0) {  // try_binary: is_nullable()
      nb = (uint8_t*)xalloc(bitmap_bytes(len));
      for (int64_t i = 0; i < len; ++i) {
        bool a = !l->validity || get_bit(l->validity, l->validity_bit_offset + i);
        bool b = !r->validity || get_bit(r->validity, r->validity_bit_offset + i);
        if (a && b) set_bit(nb, i);
      }
    }
    for (int64_t i = 0; i < len; ++i) {
      if (nb && !get_bit(nb, i)) continue;
      oi128 o;
      int32_t st = row(load(lv, i), load(rv, i), &o);
      if (st != ORC_OK) { free(nb); orc_release(out); return st; }
      memcpy((char*)out->values + i * 16, &o, 16);
    }
    attach_nulls(out, nb, len);
  }
  // .with_precision_and_scale(result_precision, result_scale)? — after the rows
  int32_t bad = ORC_OK;
  if (result_precision == 0) bad = fail(ORC_INVALID_ARGUMENT, "precision cannot be 0, has to be between [1, 38]");
  else if (result_scale > 38) bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than max 38", result_scale);
  else if (result_scale > 0 && result_scale > result_precision)
    bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than precision %d", result_scale, result_precision);
  if (bad != ORC_OK) { orc_release(out); return bad; }
  *ot = *lt;
  ot->precision = result_precision;
  ot->scale = result_scale;
  return ORC_OK;
}

// cast_decimal_to_decimal_same_type (arrow-cast/src/cast/decimal.rs:448-489) for Decimal128, native __int128
std::string fmt_decimal(const std::string& value_str, size_t precision, int scale, bool safe_decimal) {  // decimal.rs:1137-1167
  bool neg = !value_str.empty() && value_str[0] == '-';
  std::string sign = neg ? "-" : "", rest = neg ? value_str.substr(1) : value_str;
  size_t bound = safe_decimal ? std::min(precision, rest.size()) + sign.size() : value_str.size();
  std::string v = value_str.substr(0, bound);
  if (scale == 0) return v;
  if (scale < 0) return v + std::string((size_t)-scale, '0');
  if (rest.size() > (size_t)scale) return v.substr(0, v.size() - scale) + "." + v.substr(v.size() - scale);
  return sign + "0." + std::string((size_t)scale - rest.size(), '0') + rest;
}
int32_t cast_decimal128(const orc_view* in, const orc_data_type* from, const orc_data_type* to, bool safe, orc_out* out) {
  const int ip = from->precision, is = from->scale, op = to->precision, os = to->scale;
  const int64_t len = in->length;
  auto at = [&](int64_t i) { oi128 x; memcpy(&x, (const char*)in->values + i * 16, 16); return x; };
  oi128 max_v = 0;
  { oi128 t = 1; for (int i = 0; i < std::min(std::max(op, 0), 38); ++i) t *= 10; max_v = t - 1; }
  auto valid_precision = [&](oi128 v) { return op <= 38 && v <= max_v && v >= -max_v; };
  auto start = [&] {
    out->type = ORC_FIXED16;
    out->length = len;
    out->values = xalloc((size_t)len * 16);
    out->values_bytes = len * 16;
  };
  auto put = [&](int64_t i, oi128 v) { memcpy((char*)out->values + i * 16, &v, 16); };
  std::function<bool(oi128, oi128*)> f_fallible;
  bool infallible = false, clone = false, zeros = false;
  oi128 k = 1;
  if (is == os && ip <= op) clone = true;
  else if (is <= os) {
    int delta = (int8_t)(os - is);
    if (delta < 0 || delta > 38)
      return fail(ORC_CAST_ERROR, "Cannot cast to Decimal128(%d, %d). Value overflows for output scale", op, os);
    for (int i = 0; i < delta; ++i) k *= 10;
    f_fallible = [k](oi128 x, oi128* o) { return !__builtin_mul_overflow(x, k, o); };
    infallible = ((int8_t)ip + delta) <= (int8_t)op;
  } else {
    int delta = (int8_t)(is - os);
    if (delta < 0 || delta > 38) zeros = true;
    else {
      for (int i = 0; i < delta; ++i) k *= 10;
      const oi128 half = k / 2;
      f_fallible = [k, half](oi128 x, oi128* o) {
        oi128 d = x / k, r = x % k;
        if (x >= 0 && r >= half) d += 1;
        else if (x < 0 && r <= -half) d -= 1;
        *o = d;
        return true;
      };
      infallible = ((int8_t)ip - delta) < (int8_t)op;
    }
  }
  start();
  uint8_t* nb = nullptr;
  if (clone || zeros || infallible) {  // array.clone() / zeros with the nulls / array.unary(f_infallible): every slot
    for (int64_t i = 0; i < len; ++i) {
      oi128 x = at(i), o = x;
      if (zeros) o = 0;
      else if (!clone) {
        if (is <= os) o = (oi128)((unsigned __int128)x * (unsigned __int128)k);  // mul_wrapping
        else f_fallible(x, &o);
      }
      put(i, o);
    }
    nb = nulls_clone(in, len);
  } else if (safe) {  // unary_opt(f(x).filter(is_valid_decimal_precision))
    nb = (uint8_t*)xalloc(bitmap_bytes(len));
    for (int64_t i = 0; i < len; ++i) {
      if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
      oi128 o;
      if (f_fallible(at(i), &o) && valid_precision(o)) { put(i, o); set_bit(nb, i); }
    }
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
    nb = nullptr;
  } else {  // try_unary
    for (int64_t i = 0; i < len; ++i) {
      if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
      oi128 x = at(i), o;
      int32_t st = ORC_OK;
      if (!f_fallible(x, &o)) st = fail(ORC_CAST_ERROR, "Cannot cast to Decimal128(%d, %d). Overflowing on %s", op, os, i128_dbg(x).c_str());
      else if (o > max_v)
        st = fail(ORC_INVALID_ARGUMENT, "%s is too large to store in a Decimal128 of precision %d. Max is %s",
                  fmt_decimal(i128_dbg(o), op, os, false).c_str(), op, fmt_decimal(i128_dbg(max_v), op, os, true).c_str());
      else if (o < -max_v)
        st = fail(ORC_INVALID_ARGUMENT, "%s is too small to store in a Decimal128 of precision %d. Min is %s",
                  fmt_decimal(i128_dbg(o), op, os, false).c_str(), op, fmt_decimal(i128_dbg(-max_v), op, os, true).c_str());
      if (st != ORC_OK) { orc_release(out); return st; }
      put(i, o);
    }
    nb = nulls_clone(in, len);
  }
  if (nb) attach_nulls(out, nb, len);
  int32_t bad = ORC_OK;  // with_precision_and_scale(output_precision, output_scale)?
  if (op == 0) bad = fail(ORC_INVALID_ARGUMENT, "precision cannot be 0, has to be between [1, 38]");
  else if (op > 38) bad = fail(ORC_INVALID_ARGUMENT, "precision %d is greater than max 38", op);
  else if (os > 38) bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than max 38", os);
  else if (os > 0 && os > op) bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than precision %d", os, op);
  if (bad != ORC_OK) { orc_release(out); return bad; }
  return ORC_OK;
}

// cast_integer_to_decimal (arrow-cast/src/cast/mod.rs:366-443) to Decimal128
template <typename T>
int32_t cast_int_to_decimal128(const orc_view* in, const orc_data_type* to, bool safe, orc_out* out) {
  const int precision = to->precision, scale = to->scale;
  const int64_t len = in->length;
  const T* iv = (const T*)in->values;
  oi128 max_v = 0;
  { oi128 t = 1; for (int i = 0; i < std::min(std::max(precision, 0), 38); ++i) t *= 10; max_v = t - 1; }
  out->type = ORC_FIXED16;
  out->length = len;
  out->values = xalloc((size_t)len * 16);
  out->values_bytes = len * 16;
  auto put = [&](int64_t i, oi128 v) { memcpy((char*)out->values + i * 16, &v, 16); };
  auto precision_error = [&](oi128 v) {
    if (v > max_v)
      return fail(ORC_INVALID_ARGUMENT, "%s is too large to store in a Decimal128 of precision %d. Max is %s",
                  fmt_decimal(i128_dbg(v), precision, scale, false).c_str(), precision, fmt_decimal(i128_dbg(max_v), precision, scale, true).c_str());
    return fail(ORC_INVALID_ARGUMENT, "%s is too small to store in a Decimal128 of precision %d. Min is %s",
                fmt_decimal(i128_dbg(v), precision, scale, false).c_str(), precision, fmt_decimal(i128_dbg(-max_v), precision, scale, true).c_str());
  };
  std::function<int32_t(T, oi128*)> row;  // ORC_OK, or the error already set through fail()
  bool zeros = false;
  if (scale < 0) {
    T factor = 1;  // T::Native::usize_as(10).pow_checked(scale.unsigned_abs())
    bool some = true;
    for (int i = 0; i < -scale && some; ++i) some = !__builtin_mul_overflow(factor, (T)10, &factor);
    if (!some) zeros = true;
    else row = [=](T v, oi128* o) -> int32_t {
      oi128 q = (oi128)(T)(v / factor);
      if (q > max_v || q < -max_v) return precision_error(q);
      *o = q;
      return ORC_OK;
    };
  } else {
    oi128 factor = 1;
    for (int i = 0; i < scale; ++i)
      if (__builtin_mul_overflow(factor, (oi128)10, &factor)) {
        orc_release(out);
        return fail(ORC_CAST_ERROR, "Cannot cast to \"Decimal128\"(%d, %d). The scale causes overflow.", precision, scale);
      }
    row = [=](T v, oi128* o) -> int32_t {
      oi128 r;
      if (__builtin_mul_overflow((oi128)v, factor, &r))
        return fail(ORC_ARITHMETIC_OVERFLOW, "Overflow happened on: %s * %s", i128_dbg((oi128)v).c_str(), i128_dbg(factor).c_str());
      if (r > max_v || r < -max_v) return precision_error(r);
      *o = r;
      return ORC_OK;
    };
  }
  if (zeros) {  // array.unary(|_| ZERO)
    attach_nulls(out, nulls_clone(in, len), len);
  } else if (safe) {  // unary_opt
    uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
    for (int64_t i = 0; i < len; ++i) {
      if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
      oi128 o;
      if (row(iv[i], &o) == ORC_OK) { put(i, o); set_bit(nb, i); }
    }
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(len);
    out->null_count = len - count_set_bits(nb, 0, len);
  } else {  // try_unary
    for (int64_t i = 0; i < len; ++i) {
      if (in->validity && !get_bit(in->validity, in->validity_bit_offset + i)) continue;
      oi128 o;
      int32_t st = row(iv[i], &o);
      if (st != ORC_OK) { orc_release(out); return st; }
      put(i, o);
    }
    attach_nulls(out, nulls_clone(in, len), len);
  }
  int32_t bad = ORC_OK;
  if (precision == 0) bad = fail(ORC_INVALID_ARGUMENT, "precision cannot be 0, has to be between [1, 38]");
  else if (precision > 38) bad = fail(ORC_INVALID_ARGUMENT, "precision %d is greater than max 38", precision);
  else if (scale > 38) bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than max 38", scale);
  else if (scale > 0 && scale > precision) bad = fail(ORC_INVALID_ARGUMENT, "scale %d is greater than precision %d", scale, precision);
  if (bad != ORC_OK) { orc_release(out); return bad; }
  return ORC_OK;
}

int32_t arith_temporal(int op, const orc_view* l, bool l_s, const orc_data_type* lt, const orc_view* r, bool r_s,
                       const orc_data_type* rt, orc_out* out, orc_data_type* ot) {
  out_init(out);
  const int L = lt->id, R = rt->id;
  const bool add = op == 0 || op == 1, sub = op == 2 || op == 3, commutative = add || op == 4 || op == 5;
  auto logical = [](int id) { return id >= DT_DATE32; };
  if (!logical(L) && !logical(R)) {
    *ot = *lt;
    return orc_arith(op, l, l_s, r, r_s, out);
  }
  if (L == DT_DECIMAL128 && R == DT_DECIMAL128) return decimal_op(op, l, l_s, lt, r, r_s, rt, out, ot);
  const std::string ls = arith_dt_text(lt), rs = arith_dt_text(rt);
  auto nyi = [&] { return fail(ORC_NOT_YET_IMPLEMENTED, "%s %s %s: interval arithmetic is not built on the device", ls.c_str(), op_display(op), rs.c_str()); };
  if (L == DT_TIMESTAMP) {
    if (sub && R == DT_TIMESTAMP && rt->unit == lt->unit) {  // :440-443 try_op_ref!(T::Duration, .. l.sub_checked(r))
      *ot = with_unit(DT_DURATION, lt->unit);
      return orc_arith(2, l, l_s, r, r_s, out);
    }
    if (R == DT_DURATION && rt->unit == lt->unit && (add || sub)) {  // :445-452, result keeps the left zone (:536)
      *ot = *lt;
      return orc_arith(add ? 0 : 2, l, l_s, r, r_s, out);
    }
    if (R == DT_INTERVAL && (add || sub)) return nyi();
    return fail(ORC_INVALID_ARGUMENT, "Invalid timestamp arithmetic operation: %s %s %s", ls.c_str(), op_display(op), rs.c_str());
  }
  if (L == DT_DURATION && R == DT_DURATION && lt->unit == rt->unit) {
    if (add || sub) {
      *ot = *lt;
      return orc_arith(add ? 0 : 2, l, l_s, r, r_s, out);
    }
    return fail(ORC_INVALID_ARGUMENT, "Invalid duration arithmetic operation: %s %s %s", ls.c_str(), op_display(op), rs.c_str());
  }
  if (L == DT_INTERVAL && (R == DT_INTERVAL ? rt->unit == lt->unit : (R == ORC_INT64 || (R == ORC_FLOAT64 && lt->unit == 2)))) return nyi();
  if (L == DT_DATE32 || L == DT_DATE64) {
    if (sub && R == L) {
      if (L == DT_DATE64) {  // :926-930
        *ot = with_unit(DT_DURATION, U_MS);
        return orc_arith(2, l, l_s, r, r_s, out);
      }
      // :913-924 op_ref!(DurationSecondType, .., ((l as i64) - (r as i64)) * NUM_SECONDS_IN_DAY): `binary` / `unary`
      *ot = with_unit(DT_DURATION, U_S);
      const int32_t* lv = (const int32_t*)l->values;
      const int32_t* rv = (const int32_t*)r->values;
      auto f = [](int32_t a, int32_t b) { return ((int64_t)a - (int64_t)b) * 86400; };
      if (l_s != r_s) {
        const orc_view* arr = l_s ? r : l;
        const orc_view* sc = l_s ? l : r;
        const int64_t len = arr->length;
        int64_t* ov = start_out<int64_t>(arr, ORC_INT64, out);
        if (resolve_nulls(sc) != 0) {  // new_null(len)
          memset(ov, 0, (size_t)len * 8);
          out->validity = (uint8_t*)xalloc(bitmap_bytes(len));
          out->validity_bytes = (int64_t)bitmap_bytes(len);
          out->null_count = len;
          return ORC_OK;
        }
        for (int64_t i = 0; i < len; ++i) ov[i] = l_s ? f(lv[0], rv[i]) : f(lv[i], rv[0]);
        attach_nulls(out, nulls_clone(arr, len), len);
        return ORC_OK;
      }
      if (l->length != r->length) return fail(ORC_COMPUTE_ERROR, "Cannot perform binary operation on arrays of different length");
      const int64_t len = l->length;
      int64_t* ov = start_out<int64_t>(l, ORC_INT64, out);
      for (int64_t i = 0; i < len; ++i) ov[i] = f(lv[i], rv[i]);
      if (l->validity || r->validity) {  // NullBuffer::union: presence-based
        uint8_t* nb = (uint8_t*)xalloc(bitmap_bytes(len));
        for (int64_t i = 0; i < len; ++i) {
          bool a = !l->validity || get_bit(l->validity, l->validity_bit_offset + i);
          bool b = !r->validity || get_bit(r->validity, r->validity_bit_offset + i);
          if (a && b) set_bit(nb, i);
        }
        attach_nulls(out, nb, len);
      }
      return ORC_OK;
    }
    if (R == DT_INTERVAL && (add || sub)) return nyi();
    return fail(ORC_INVALID_ARGUMENT, "Invalid date arithmetic operation: %s %s %s", ls.c_str(), op_display(op), rs.c_str());
  }
  if ((L == DT_DURATION || L == DT_INTERVAL) && (R == DT_DATE32 || R == DT_DATE64 || R == DT_TIMESTAMP) && commutative)
    return arith_temporal(op, r, r_s, rt, l, l_s, lt, out, ot);  // :263-265
  if (((L == ORC_INT64 && R == DT_INTERVAL) || (L == ORC_FLOAT64 && R == DT_INTERVAL && rt->unit == 2)) && op == 4) return nyi();
  return fail(ORC_INVALID_ARGUMENT, "Invalid arithmetic operation: %s %s %s", ls.c_str(), op_display(op), rs.c_str());
}
}  // namespace

extern "C" {

int32_t orc_cast_with_types(const orc_view* in, const orc_data_type* from, const orc_data_type* to, int32_t safe,
                            orc_out* out) {
  if (from->id == 39 && to->id == 39) {
    out_init(out);
    return cast_decimal128(in, from, to, safe != 0, out);
  }
  if (is_integer(from->id) && to->id == 39) {
    out_init(out);
    switch (from->id) {
      case ORC_INT8: return cast_int_to_decimal128<int8_t>(in, to, safe != 0, out);
      case ORC_INT16: return cast_int_to_decimal128<int16_t>(in, to, safe != 0, out);
      case ORC_INT32: return cast_int_to_decimal128<int32_t>(in, to, safe != 0, out);
      case ORC_INT64: return cast_int_to_decimal128<int64_t>(in, to, safe != 0, out);
      case ORC_UINT8: return cast_int_to_decimal128<uint8_t>(in, to, safe != 0, out);
      case ORC_UINT16: return cast_int_to_decimal128<uint16_t>(in, to, safe != 0, out);
      case ORC_UINT32: return cast_int_to_decimal128<uint32_t>(in, to, safe != 0, out);
      default: return cast_int_to_decimal128<uint64_t>(in, to, safe != 0, out);
    }
  }
  return cast_temporal(in, from, to, safe != 0, out);
}

int32_t orc_arith_with_types(int32_t op, const orc_view* lhs, int32_t lhs_scalar, const orc_data_type* lhs_type,
                             const orc_view* rhs, int32_t rhs_scalar, const orc_data_type* rhs_type, orc_out* out,
                             orc_data_type* out_type) {
  return arith_temporal(op, lhs, lhs_scalar != 0, lhs_type, rhs, rhs_scalar != 0, rhs_type, out, out_type);
}

int32_t orc_aggregate(int32_t op, const orc_view* a, int32_t vector_bytes, orc_scalar* out) {
  memset(out, 0, sizeof *out);
  out->type = a->type;
  const int vb = vector_bytes ? vector_bytes : 16;
  switch (a->type) {
    case ORC_BOOL: return aggregate_boolean(op, a, out);
    case ORC_INT8: return aggregate_dispatch<int8_t>(op, a, vb, out);
    case ORC_INT16: return aggregate_dispatch<int16_t>(op, a, vb, out);
    case ORC_INT32: return aggregate_dispatch<int32_t>(op, a, vb, out);
    case ORC_INT64: return aggregate_dispatch<int64_t>(op, a, vb, out);
    case ORC_UINT8: return aggregate_dispatch<uint8_t>(op, a, vb, out);
    case ORC_UINT16: return aggregate_dispatch<uint16_t>(op, a, vb, out);
    case ORC_UINT32: return aggregate_dispatch<uint32_t>(op, a, vb, out);
    case ORC_UINT64: return aggregate_dispatch<uint64_t>(op, a, vb, out);
    case ORC_FLOAT32: return aggregate_dispatch<float>(op, a, vb, out);
    case ORC_FLOAT64: return aggregate_dispatch<double>(op, a, vb, out);
  }
  return fail(ORC_NOT_YET_IMPLEMENTED, "aggregate of %s", type_name(a->type));
}

// ---- parquet RowSelection on masks
int32_t orc_selection_and_then(const orc_view* m, const orc_view* o, orc_out* out) {  // and_then_masks
  out_init(out);
  const uint8_t* mb = (const uint8_t*)m->values;
  const uint8_t* ob = (const uint8_t*)o->values;
  const int64_t selected = count_set_bits(mb, m->values_bit_offset, m->length);
  if (o->length < selected) return fail(ORC_PANIC, "selection contains less than the number of selected rows");
  if (o->length > selected) return fail(ORC_PANIC, "selection exceeds the number of selected rows");
  out->type = ORC_BOOL;
  out->length = m->length;
  uint8_t* r = (uint8_t*)xalloc(bitmap_bytes(m->length));
  out->values = r;
  out->values_bytes = (int64_t)bitmap_bytes(m->length);
  int64_t ordinal = 0;  // walk the set positions of `mask`, consuming one bit of `other` each
  for (int64_t i = 0; i < m->length; ++i)
    if (get_bit(mb, m->values_bit_offset + i)) {
      if (get_bit(ob, o->values_bit_offset + ordinal)) set_bit(r, i);
      ++ordinal;
    }
  return ORC_OK;
}

int32_t orc_selection_combine(int32_t op, const orc_view* l, const orc_view* r, orc_out* out) {
  out_init(out);
  const orc_view* longer = l->length > r->length ? l : r;  // combine_unequal_length_masks :319-349
  const orc_view* shorter = longer == l ? r : l;
  out->type = ORC_BOOL;
  out->length = longer->length;
  uint8_t* res = (uint8_t*)xalloc(bitmap_bytes(longer->length));
  out->values = res;
  out->values_bytes = (int64_t)bitmap_bytes(longer->length);
  const uint8_t* lb = (const uint8_t*)longer->values;
  const uint8_t* sb = (const uint8_t*)shorter->values;
  for (int64_t i = 0; i < longer->length; ++i) {
    bool v = get_bit(lb, longer->values_bit_offset + i);
    if (i < shorter->length) {
      const bool w = get_bit(sb, shorter->values_bit_offset + i);
      v = op == 0 ? (v && w) : (v || w);
    }
    if (v) set_bit(res, i);
  }
  return ORC_OK;
}

int64_t orc_find_nth_set_bit(const uint8_t* bits, int64_t off, int64_t len, int64_t start, int64_t n) {
  if (n == 0) return start;
  for (int64_t i = start; i < len; ++i)
    if (get_bit(bits, off + i) && --n == 0) return i + 1;
  return len;
}


// &[u8] Ord (sort_bytes / bytes_rank): bytewise lexicographic, a proper prefix first
struct BytesCol {
  const orc_view* a;
  bool large;
  explicit BytesCol(const orc_view* v) : a(v), large(v->type == ORC_LARGE_UTF8) {}
  int64_t off(int64_t i) const { return large ? ((const int64_t*)a->offsets)[i] : (int64_t)((const int32_t*)a->offsets)[i]; }
  int cmp(uint32_t x, uint32_t y) const {
    const int64_t ax = off(x), nx = off(x + 1) - ax, ay = off(y), ny = off(y + 1) - ay;
    const int c = memcmp((const uint8_t*)a->values + ax, (const uint8_t*)a->values + ay, (size_t)std::min(nx, ny));
    return c ? (c < 0 ? -1 : 1) : (nx < ny ? -1 : nx > ny ? 1 : 0);
  }
};

// sort_to_indices (arrow-ord/src/sort.rs:276-300): partition_validity :193-255, sort_primitive :341-352,
// sort_boolean :325-339, sort_impl :639-672
int32_t orc_sort_to_indices(const orc_view* a, int32_t desc, int32_t nulls_first, int64_t limit, orc_out* out) {
  out_init(out);
  out->type = ORC_UINT32;
  const int64_t n = a->length;
  if (n == 0 || limit == 0) return ORC_OK;
  std::vector<uint32_t> valid, nulls;
  const bool has_nulls = a->validity && resolve_nulls(a) > 0;
  for (int64_t i = 0; i < n; ++i) {
    if (has_nulls && !get_bit(a->validity, a->validity_bit_offset + i)) nulls.push_back((uint32_t)i);
    else valid.push_back((uint32_t)i);
  }
  switch (a->type) {
    case ORC_BOOL: {
      const uint8_t* b = (const uint8_t*)a->values;
      const int64_t off = a->values_bit_offset;
      auto val = [b, off](uint32_t i) { return get_bit(b, off + i); };
      if (!desc) std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return val(x) < val(y); });
      else std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return val(y) < val(x); });
      break;
    }
    case ORC_INT8: sort_valid_typed<int8_t>(a, valid, desc); break;
    case ORC_INT16: sort_valid_typed<int16_t>(a, valid, desc); break;
    case ORC_INT32: sort_valid_typed<int32_t>(a, valid, desc); break;
    case ORC_INT64: sort_valid_typed<int64_t>(a, valid, desc); break;
    case ORC_UINT8: sort_valid_typed<uint8_t>(a, valid, desc); break;
    case ORC_UINT16: sort_valid_typed<uint16_t>(a, valid, desc); break;
    case ORC_UINT32: sort_valid_typed<uint32_t>(a, valid, desc); break;
    case ORC_UINT64: sort_valid_typed<uint64_t>(a, valid, desc); break;
    case ORC_FLOAT32: sort_valid_typed<float>(a, valid, desc); break;
    case ORC_FLOAT64: sort_valid_typed<double>(a, valid, desc); break;
    case ORC_FLOAT16: {  // total order on the 16 raw bits
      const uint16_t* v = (const uint16_t*)a->values;
      auto key = [v](uint32_t i) { const uint16_t b = v[i]; return (uint16_t)((b & 0x8000) ? ~b : (b ^ 0x8000)); };
      if (!desc) std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return key(x) < key(y); });
      else std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return key(y) < key(x); });
      break;
    }
    case ORC_UTF8:
    case ORC_LARGE_UTF8: {  // sort_bytes (sort.rs): values.cmp as &[u8]
      const BytesCol bc(a);
      if (!desc) std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return bc.cmp(x, y) < 0; });
      else std::stable_sort(valid.begin(), valid.end(), [&](uint32_t x, uint32_t y) { return bc.cmp(y, x) < 0; });
      break;
    }
    default: return fail(ORC_COMPUTE_ERROR, "Sort not supported for data type %s", type_name(a->type));
  }
  const int64_t len = (int64_t)valid.size() + (int64_t)nulls.size();
  const int64_t lim = limit < 0 ? len : std::min(limit, len);
  uint32_t* o = (uint32_t*)xalloc((size_t)std::max<int64_t>(lim, 1) * 4);
  int64_t k = 0;
  if (nulls_first) {
    for (size_t i = 0; i < nulls.size() && k < lim; ++i) o[k++] = nulls[i];
    for (size_t i = 0; i < valid.size() && k < lim; ++i) o[k++] = valid[i];
  } else {
    for (size_t i = 0; i < valid.size() && k < lim; ++i) o[k++] = valid[i];
    for (size_t i = 0; i < nulls.size() && k < lim; ++i) o[k++] = nulls[i];
  }
  out->length = lim;
  out->values = o;
  out->values_bytes = std::max<int64_t>(lim, 1) * 4;
  return ORC_OK;
}

// rank (arrow-ord/src/rank.rs:58-160): primitive_rank :72-87 + rank_impl :119-160, followed literally — sort the
// valid (value, row) pairs, reverse when descending, walk backwards handing out `valid_rank` and lowering it by the size
// of each finished run of equal values; nulls get `null_rank`.  boolean_rank :182-245 computes the same from three
// counts and is restated separately.
int32_t orc_rank(const orc_view* a, int32_t desc, int32_t nulls_first, orc_out* out) {
  out_init(out);
  out->type = ORC_UINT32;
  const int64_t len = a->length;
  if (len == 0) return ORC_OK;
  const bool has_nulls = a->validity && resolve_nulls(a) > 0;  // nulls.filter(|n| n.null_count() > 0)
  uint32_t* o = (uint32_t*)xalloc((size_t)len * 4);
  out->length = len;
  out->values = o;
  out->values_bytes = len * 4;
  if (a->type == ORC_BOOL) {
    const uint8_t* b = (const uint8_t*)a->values;
    uint32_t null_count = 0, true_count = 0;
    for (int64_t i = 0; i < len; ++i) {
      const bool valid = !has_nulls || get_bit(a->validity, a->validity_bit_offset + i);
      if (!valid) ++null_count;
      else if (get_bit(b, a->values_bit_offset + i)) ++true_count;
    }
    const uint32_t false_count = (uint32_t)len - null_count - true_count;
    uint32_t r[3];  // [false, true, null]
    if (desc && nulls_first) r[0] = null_count + true_count + false_count, r[1] = null_count + true_count, r[2] = null_count;
    else if (desc) r[0] = true_count + false_count, r[1] = true_count, r[2] = true_count + false_count + null_count;
    else if (nulls_first) r[0] = null_count + false_count, r[1] = null_count + false_count + true_count, r[2] = null_count;
    else r[0] = false_count, r[1] = false_count + true_count, r[2] = false_count + true_count + null_count;
    for (int64_t i = 0; i < len; ++i) {
      const bool valid = !has_nulls || get_bit(a->validity, a->validity_bit_offset + i);
      o[i] = r[!valid ? 2 : (get_bit(b, a->values_bit_offset + i) ? 1 : 0)];
    }
    return ORC_OK;
  }
  std::vector<uint32_t> valid;
  for (int64_t i = 0; i < len; ++i)
    if (!has_nulls || get_bit(a->validity, a->validity_bit_offset + i)) valid.push_back((uint32_t)i);
  std::function<bool(uint32_t, uint32_t)> lt, eq;
  switch (a->type) {
#define ORC_RANK_CASE(TAG, T)                                                          \
  case TAG: {                                                                          \
    const T* v = (const T*)a->values;                                                  \
    lt = [v](uint32_t x, uint32_t y) { return is_lt<T>(v[x], v[y]); };                 \
    eq = [v](uint32_t x, uint32_t y) { return !is_lt<T>(v[x], v[y]) && !is_lt<T>(v[y], v[x]); }; \
    break;                                                                             \
  }
    ORC_RANK_CASE(ORC_INT8, int8_t)
    ORC_RANK_CASE(ORC_INT16, int16_t)
    ORC_RANK_CASE(ORC_INT32, int32_t)
    ORC_RANK_CASE(ORC_INT64, int64_t)
    ORC_RANK_CASE(ORC_UINT8, uint8_t)
    ORC_RANK_CASE(ORC_UINT16, uint16_t)
    ORC_RANK_CASE(ORC_UINT32, uint32_t)
    ORC_RANK_CASE(ORC_UINT64, uint64_t)
    ORC_RANK_CASE(ORC_FLOAT32, float)
    ORC_RANK_CASE(ORC_FLOAT64, double)
#undef ORC_RANK_CASE
    case ORC_UTF8:
    case ORC_LARGE_UTF8: {  // bytes_rank :90-101
      const BytesCol bc(a);
      lt = [bc](uint32_t x, uint32_t y) { return bc.cmp(x, y) < 0; };
      eq = [bc](uint32_t x, uint32_t y) { return bc.cmp(x, y) == 0; };
      break;
    }
    default:
      free(o);
      out_init(out);
      return fail(ORC_COMPUTE_ERROR, "%s not supported in rank", type_name(a->type));
  }
  std::sort(valid.begin(), valid.end(), lt);  // sort_unstable_by: ties are merged below, their order does not matter
  if (desc) std::reverse(valid.begin(), valid.end());
  uint32_t valid_rank = nulls_first ? (uint32_t)len : (uint32_t)valid.size();
  const uint32_t null_rank = nulls_first ? (uint32_t)(len - (int64_t)valid.size()) : (uint32_t)len;
  for (int64_t i = 0; i < len; ++i) o[i] = null_rank;
  if (!valid.empty()) o[valid.back()] = valid_rank;
  uint32_t count = 1;
  for (size_t k = valid.size(); k-- > 1;) {  // windows(2).rev(): w = [valid[k-1], valid[k]]
    if (eq(valid[k - 1], valid[k])) {
      ++count;
    } else {
      valid_rank -= count;
      count = 1;
    }
    o[valid[k - 1]] = valid_rank;
  }
  return ORC_OK;
}

int32_t orc_lexsort_to_indices(int32_t n_cols, const orc_view* cols, const int32_t* desc, const int32_t* nulls_first,
                               int64_t limit, orc_out* out) {
  out_init(out);
  out->type = ORC_UINT32;
  if (n_cols <= 0) return fail(ORC_INVALID_ARGUMENT, "Sort requires at least one column");
  if (n_cols == 1) return orc_sort_to_indices(&cols[0], desc[0], nulls_first[0], limit, out);
  const int64_t n = cols[0].length;
  for (int c = 1; c < n_cols; ++c)
    if (cols[c].length != n) return fail(ORC_COMPUTE_ERROR, "lexical sort columns have different row counts");
  const int64_t lim = limit < 0 ? n : std::min(limit, n);
  if (lim == 0) return ORC_OK;
  // per column: -1 / 0 / +1 on two valid rows (ascending), as T::Native::compare
  std::vector<std::function<int(uint32_t, uint32_t)>> cmp((size_t)n_cols);
  for (int c = 0; c < n_cols; ++c) {
    const orc_view* a = &cols[c];
    switch (a->type) {
#define ORC_CMP_CASE(TAG, T)                                                                          \
  case TAG: {                                                                                         \
    const T* v = (const T*)a->values;                                                                 \
    cmp[c] = [v](uint32_t x, uint32_t y) { return is_lt<T>(v[x], v[y]) ? -1 : is_lt<T>(v[y], v[x]) ? 1 : 0; }; \
    break;                                                                                            \
  }
      ORC_CMP_CASE(ORC_INT8, int8_t)
      ORC_CMP_CASE(ORC_INT16, int16_t)
      ORC_CMP_CASE(ORC_INT32, int32_t)
      ORC_CMP_CASE(ORC_INT64, int64_t)
      ORC_CMP_CASE(ORC_UINT8, uint8_t)
      ORC_CMP_CASE(ORC_UINT16, uint16_t)
      ORC_CMP_CASE(ORC_UINT32, uint32_t)
      ORC_CMP_CASE(ORC_UINT64, uint64_t)
      ORC_CMP_CASE(ORC_FLOAT32, float)
      ORC_CMP_CASE(ORC_FLOAT64, double)
#undef ORC_CMP_CASE
      case ORC_BOOL: {
        const uint8_t* b = (const uint8_t*)a->values;
        const int64_t off = a->values_bit_offset;
        cmp[c] = [b, off](uint32_t x, uint32_t y) { return (int)get_bit(b, off + x) - (int)get_bit(b, off + y); };
        break;
      }
      case ORC_UTF8:
      case ORC_LARGE_UTF8: {
        const BytesCol bc(a);
        cmp[c] = [bc](uint32_t x, uint32_t y) { return bc.cmp(x, y); };
        break;
      }
      default: return fail(ORC_COMPUTE_ERROR, "Sort not supported for data type %s", type_name(a->type));
    }
  }
  std::vector<bool> has_nulls((size_t)n_cols);
  for (int c = 0; c < n_cols; ++c) has_nulls[c] = cols[c].validity && resolve_nulls(&cols[c]) > 0;
  auto less = [&](uint32_t x, uint32_t y) {
    for (int c = 0; c < n_cols; ++c) {
      const orc_view* a = &cols[c];
      const bool vx = !has_nulls[c] || get_bit(a->validity, a->validity_bit_offset + x);
      const bool vy = !has_nulls[c] || get_bit(a->validity, a->validity_bit_offset + y);
      int r;
      if (!vx && !vy) r = 0;
      else if (!vx) r = nulls_first[c] ? -1 : 1;
      else if (!vy) r = nulls_first[c] ? 1 : -1;
      else r = desc[c] ? -cmp[c](x, y) : cmp[c](x, y);
      if (r) return r < 0;
    }
    return false;
  };
  std::vector<uint32_t> idx((size_t)n);
  for (int64_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
  std::stable_sort(idx.begin(), idx.end(), less);
  uint32_t* o = (uint32_t*)xalloc((size_t)lim * 4);
  memcpy(o, idx.data(), (size_t)lim * 4);
  out->length = lim;
  out->values = o;
  out->values_bytes = lim * 4;
  return ORC_OK;
}

// zip (arrow-select/src/zip.rs:99-200): argument checks :115-140; nulls in the mask select `falsy`
// (maybe_prep_null_mask_filter); the output carries a null buffer iff an input has nulls (MutableArrayData)
int32_t orc_zip(const orc_view* mask, const orc_view* t, int32_t ts, const orc_view* f, int32_t fs, orc_out* out) {
  out_init(out);
  if (t->type != f->type) return fail(ORC_INVALID_ARGUMENT, "arguments need to have the same data type");
  if (ts && t->length != 1) return fail(ORC_INVALID_ARGUMENT, "scalar arrays must have 1 element");
  if (!ts && t->length != mask->length) return fail(ORC_INVALID_ARGUMENT, "all arrays should have the same length");
  if (fs && f->length != 1) return fail(ORC_INVALID_ARGUMENT, "scalar arrays must have 1 element");
  if (!fs && f->length != mask->length) return fail(ORC_INVALID_ARGUMENT, "all arrays should have the same length");
  const int w = type_width(t->type);
  if (w < 0) return fail(ORC_NOT_YET_IMPLEMENTED, "zip of %s", type_name(t->type));
  const int64_t n = mask->length;
  out->type = t->type;
  out->length = n;
  if (n == 0) return ORC_OK;
  const bool any_nulls = resolve_nulls(t) > 0 || resolve_nulls(f) > 0;
  const size_t vbytes = w ? (size_t)n * w : bitmap_bytes(n);
  out->values = xalloc(vbytes);
  out->values_bytes = (int64_t)vbytes;
  uint8_t* nb = any_nulls ? (uint8_t*)xalloc(bitmap_bytes(n)) : nullptr;
  const uint8_t* mv = (const uint8_t*)mask->values;
  int64_t valid_count = 0;
  for (int64_t i = 0; i < n; ++i) {
    bool m = get_bit(mv, mask->values_bit_offset + i);
    if (mask->validity && !get_bit(mask->validity, mask->validity_bit_offset + i)) m = false;
    const orc_view* src = m ? t : f;
    const int64_t j = (m ? ts : fs) ? 0 : i;
    if (w) memcpy((char*)out->values + (size_t)i * w, (const char*)src->values + (size_t)j * w, (size_t)w);
    else if (get_bit((const uint8_t*)src->values, src->values_bit_offset + j)) set_bit((uint8_t*)out->values, i);
    const bool valid = !src->validity || get_bit(src->validity, src->validity_bit_offset + j);
    if (nb && valid) set_bit(nb, i);
    valid_count += valid;
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(n);
    out->null_count = n - valid_count;
  }
  return ORC_OK;
}

int32_t orc_interleave(int32_t n, const orc_view* arrays, const uint32_t* ai, const uint32_t* ri, int64_t m, orc_out* out) {
  out_init(out);
  if (n <= 0) return fail(ORC_INVALID_ARGUMENT, "interleave requires input of at least one array");
  const int32_t t = arrays[0].type;
  for (int i = 1; i < n; ++i)
    if (arrays[i].type != t)
      return fail(ORC_INVALID_ARGUMENT, "It is not possible to interleave arrays of different data types (%s and %s)",
                  type_name(t), type_name(arrays[i].type));
  const int w = type_width(t);
  if (w < 0) return fail(ORC_NOT_YET_IMPLEMENTED, "interleave of %s", type_name(t));
  out->type = t;
  out->length = m;
  if (m == 0) return ORC_OK;
  bool any_nulls = false;
  for (int i = 0; i < n; ++i) any_nulls |= arrays[i].validity && resolve_nulls(&arrays[i]) > 0;
  const size_t vbytes = w ? (size_t)m * w : bitmap_bytes(m);
  out->values = xalloc(vbytes);
  out->values_bytes = (int64_t)vbytes;
  uint8_t* nb = any_nulls ? (uint8_t*)xalloc(bitmap_bytes(m)) : nullptr;
  int64_t valid_count = 0;
  for (int64_t i = 0; i < m; ++i) {
    if (ai[i] >= (uint32_t)n) {
      free(nb);
      orc_release(out);
      return fail(ORC_PANIC, "index out of bounds: the len is %d but the index is %u", n, ai[i]);
    }
    const orc_view* s = &arrays[ai[i]];
    if ((int64_t)ri[i] >= s->length) {
      free(nb);
      orc_release(out);
      return fail(ORC_PANIC, "index out of bounds: the len is %lld but the index is %u", (long long)s->length, ri[i]);
    }
    if (w) memcpy((char*)out->values + (size_t)i * w, (const char*)s->values + (size_t)ri[i] * w, (size_t)w);
    else if (get_bit((const uint8_t*)s->values, s->values_bit_offset + ri[i])) set_bit((uint8_t*)out->values, i);
    const bool valid = !s->validity || get_bit(s->validity, s->validity_bit_offset + ri[i]);
    if (nb && valid) set_bit(nb, i);
    valid_count += valid;
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(m);
    out->null_count = m - valid_count;
  }
  return ORC_OK;
}

int32_t orc_bitwise(int32_t op, const orc_view* l, int32_t l_s, const orc_view* r, int32_t r_s, orc_out* out) {
  out_init(out);
  if (op != 14 && l->type != r->type)
    return fail(ORC_INVALID_ARGUMENT, "Invalid arithmetic operation: %s & %s", type_name(l->type), type_name(r->type));
  switch (l->type) {
    case ORC_INT8: return bitwise_typed<int8_t>(op, l, l_s, r, r_s, out);
    case ORC_INT16: return bitwise_typed<int16_t>(op, l, l_s, r, r_s, out);
    case ORC_INT32: return bitwise_typed<int32_t>(op, l, l_s, r, r_s, out);
    case ORC_INT64: return bitwise_typed<int64_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT8: return bitwise_typed<uint8_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT16: return bitwise_typed<uint16_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT32: return bitwise_typed<uint32_t>(op, l, l_s, r, r_s, out);
    case ORC_UINT64: return bitwise_typed<uint64_t>(op, l, l_s, r, r_s, out);
  }
  return fail(ORC_INVALID_ARGUMENT, "Invalid arithmetic operation: bitwise on %s", type_name(l->type));
}

// concat for primitives / booleans (arrow-select/src/concat.rs:334-343, :495)
int32_t orc_concat(int32_t n, const orc_view* pieces, orc_out* out) {
  out_init(out);
  if (n <= 0) return fail(ORC_INVALID_ARGUMENT, "concat requires input of at least one array");
  int32_t t = pieces[0].type;
  int w = type_width(t);
  const bool is_str = t == ORC_UTF8 || t == ORC_LARGE_UTF8;
  if (w < 0 && !is_str) return fail(ORC_NOT_YET_IMPLEMENTED, "concat not supported for type %s", type_name(t));
  int64_t total = 0;
  bool any_valid_buf = false;
  for (int i = 1; i < n; ++i) {
    if (pieces[i].type == t) continue;
    // concat.rs:505-535: up to 10 unique data types in order of appearance, ", ..." once an 11th shows up
    std::string msg = std::string("It is not possible to concatenate arrays of different data types (") + type_name(t);
    std::vector<int32_t> seen{t};
    for (int j = 0; j < n; ++j) {
      const bool unique = std::find(seen.begin(), seen.end(), pieces[j].type) == seen.end();
      if (unique) seen.push_back(pieces[j].type);
      if (seen.size() == 11) {
        msg += ", ...";
        break;
      }
      if (unique) msg += std::string(", ") + type_name(pieces[j].type);
    }
    msg += ").";
    return fail(ORC_INVALID_ARGUMENT, "%s", msg.c_str());
  }
  for (int i = 0; i < n; ++i) {
    total += pieces[i].length;
    // NullBufferBuilder: a buffer materialises only if some piece has nulls
    if (pieces[i].validity && resolve_nulls(&pieces[i]) > 0) any_valid_buf = true;
  }
  out->type = t;
  out->length = total;
  if (is_str) {  // concat_bytes (concat.rs:355-368) -> GenericByteBuilder::append_array (generic_bytes_builder.rs:169-206)
    if (total == 0) return ORC_OK;
    const int ow = t == ORC_UTF8 ? 4 : 8;
    auto off = [&](const orc_view* p, int64_t i) -> int64_t {
      return ow == 4 ? (int64_t)((const int32_t*)p->offsets)[i] : ((const int64_t*)p->offsets)[i];
    };
    int64_t bytes = 0;
    for (int i = 0; i < n; ++i) {
      if (pieces[i].length == 0) continue;
      bytes += off(&pieces[i], pieces[i].length) - off(&pieces[i], 0);
      if (ow == 4 && bytes > INT32_MAX) return fail(ORC_OFFSET_OVERFLOW_ERROR, "%lld", (long long)bytes);
    }
    out->offsets = xalloc((size_t)(total + 1) * ow);
    out->offsets_bytes = (total + 1) * ow;
    out->values = xalloc((size_t)(bytes ? bytes : 8));
    out->values_bytes = bytes ? bytes : 8;
    uint8_t* nb = any_valid_buf ? (uint8_t*)xalloc(bitmap_bytes(total)) : nullptr;
    int64_t pos = 0, base = 0;
    for (int i = 0; i < n; ++i) {
      const orc_view* p = &pieces[i];
      if (p->length == 0) continue;
      const int64_t first = off(p, 0);
      for (int64_t j = 0; j <= p->length; ++j) {
        const int64_t v = off(p, j) - first + base;
        if (ow == 4) ((int32_t*)out->offsets)[pos + j] = (int32_t)v;
        else ((int64_t*)out->offsets)[pos + j] = v;
      }
      memcpy((char*)out->values + base, (const char*)p->values + first, (size_t)(off(p, p->length) - first));
      if (nb) {
        if (p->validity) copy_bits(nb, pos, p->validity, p->validity_bit_offset, p->length);
        else for (int64_t j = 0; j < p->length; ++j) set_bit(nb, pos + j);
      }
      pos += p->length;
      base += off(p, p->length) - first;
    }
    if (nb) {
      out->validity = nb;
      out->validity_bytes = (int64_t)bitmap_bytes(total);
      out->null_count = total - count_set_bits(nb, 0, total);
    }
    return ORC_OK;
  }
  size_t vbytes = w ? (size_t)total * w : bitmap_bytes(total);
  out->values = xalloc(vbytes);
  out->values_bytes = (int64_t)vbytes;
  uint8_t* nb = any_valid_buf ? (uint8_t*)xalloc(bitmap_bytes(total)) : nullptr;
  int64_t pos = 0;
  for (int i = 0; i < n; ++i) {
    const orc_view* p = &pieces[i];
    if (w) memcpy((char*)out->values + (size_t)pos * w, p->values, (size_t)p->length * w);
    else copy_bits((uint8_t*)out->values, pos, (const uint8_t*)p->values, p->values_bit_offset, p->length);
    if (nb) {
      if (p->validity) copy_bits(nb, pos, p->validity, p->validity_bit_offset, p->length);
      else for (int64_t j = 0; j < p->length; ++j) set_bit(nb, pos + j);
    }
    pos += p->length;
  }
  if (nb) {
    out->validity = nb;
    out->validity_bytes = (int64_t)bitmap_bytes(total);
    out->null_count = total - count_set_bits(nb, 0, total);
  }
  return ORC_OK;
}

int32_t orc_format_f64(double v, char* buf) { return format_float<double>(v, buf, 16, -5); }
int32_t orc_format_f32(float v, char* buf) { return format_float<float>(v, buf, 13, -6); }

void orc_gen_uniform_i64(int64_t* dst, int64_t n, uint64_t seed, int64_t lo, int64_t hi, int64_t row0) {
  uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t r = splitmix64(seed, (uint64_t)(row0 + i));
    dst[i] = range ? (int64_t)((uint64_t)lo + (uint64_t)(((unsigned __int128)r * range) >> 64)) : (int64_t)r;
  }
}
void orc_gen_uniform_i32(int32_t* dst, int64_t n, uint64_t seed, int64_t row0) {
  for (int64_t i = 0; i < n; ++i) dst[i] = (int32_t)(uint32_t)splitmix64(seed, (uint64_t)(row0 + i));
}
void orc_gen_uniform_u32(uint32_t* dst, int64_t n, uint64_t seed, uint32_t bound, int64_t row0) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t r = (uint32_t)(splitmix64(seed, (uint64_t)(row0 + i)) >> 32);
    dst[i] = bound ? (uint32_t)(((uint64_t)r * bound) >> 32) : r;
  }
}
void orc_gen_uniform_f64(double* dst, int64_t n, uint64_t seed, double lo, double hi, int64_t row0) {
  double span = hi - lo;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t r = splitmix64(seed, (uint64_t)(row0 + i));
    double u = (double)(r >> 11) * 0x1.0p-53;
    dst[i] = std::fma(u, span, lo);
  }
}
void orc_gen_bernoulli_bits(uint8_t* dst, int64_t n, uint64_t seed, double p_true, int64_t row0) {
  double p = p_true < 0 ? 0 : (p_true > 1 ? 1 : p_true);
  uint64_t threshold = (uint64_t)(p * 9007199254740992.0);
  memset(dst, 0, bitmap_bytes(n));
  for (int64_t i = 0; i < n; ++i)
    if ((splitmix64(seed, (uint64_t)(row0 + i)) >> 11) < threshold) set_bit(dst, i);
}
void orc_zero_null_slots(void* values, int32_t w, const uint8_t* validity, int64_t n) {
  if (!validity) return;
  for (int64_t i = 0; i < n; ++i)
    if (!get_bit(validity, i)) memset((char*)values + (size_t)i * w, 0, (size_t)w);
}

}  // extern "C"
