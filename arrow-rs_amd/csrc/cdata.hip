// Arrow C Data Interface <-> device arrays.
//
// Reference: arrow-array/src/ffi.rs:231-254 (to_ffi / from_ffi), :273-337 (ImportedArrowArray::
// consume), :382-470 (buffers / buffer_len), arrow-data/src/ffi.rs:99-212 (align_nulls,
// FFI_ArrowArray::new), arrow-schema/src/ffi.rs:492-700 (format string -> DataType), :779-856
// (DataType -> format string).  The reference imports zero-copy because producer and consumer
// share host memory; here the consumer's arrays live in HBM, so import is one H2D copy per buffer
// of exactly the rows [offset, offset+length) and export one D2H copy per buffer.  Host code only:
// the single device-side step (re-aligning a bitmap to bit offset 0 on export) reuses ah_bitmap_op.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace {

bool parse_int(const char* s, const char* e, long* out) {
  if (s == e) return false;
  char* end = nullptr;
  std::string tmp(s, e);
  long v = strtol(tmp.c_str(), &end, 10);
  if (*end) return false;
  *out = v;
  return true;
}

// 0 = ok, 1 = unknown format (CDataInterface), 2 = known to the reference, no kernels here
int format_to_type(const char* f, ah_type* out, std::string* why) {
  const std::string s(f);
  static const struct { const char* fmt; ah_type t; } fixed[] = {
      {"b", AH_BOOL},      {"c", AH_INT8},     {"C", AH_UINT8},   {"s", AH_INT16},   {"S", AH_UINT16},
      {"i", AH_INT32},     {"I", AH_UINT32},   {"l", AH_INT64},   {"L", AH_UINT64},  {"e", AH_FLOAT16},
      {"f", AH_FLOAT32},   {"g", AH_FLOAT64},  {"u", AH_UTF8},    {"U", AH_LARGE_UTF8},
      {"z", AH_UTF8},      {"Z", AH_LARGE_UTF8},  // Binary / LargeBinary: same buffers, bytes are never decoded
      {"tdD", AH_INT32},   {"tdm", AH_INT64},  {"tts", AH_INT32}, {"ttm", AH_INT32}, {"ttu", AH_INT64},
      {"ttn", AH_INT64},   {"tDs", AH_INT64},  {"tDm", AH_INT64}, {"tDu", AH_INT64}, {"tDn", AH_INT64},
      {"tiM", AH_INT32},   {"tiD", AH_INT64},  {"tin", AH_FIXED16}};
  for (auto& e : fixed)
    if (s == e.fmt) {
      *out = e.t;
      return 0;
    }
  if (s == "n" || s == "vu" || s == "vz" || (!s.empty() && s[0] == '+')) {
    *why = "the C Data layout \"" + s + "\" (null / view / nested) has no device kernels";
    return 2;
  }
  const size_t colon = s.find(':');
  if (colon != std::string::npos) {
    const std::string head = s.substr(0, colon), tail = s.substr(colon + 1);
    if (head == "tss" || head == "tsm" || head == "tsu" || head == "tsn") {  // ffi.rs:680-687
      *out = AH_INT64;
      return 0;
    }
    if (head == "d") {  // ffi.rs:578-615: "d:p,s" or "d:p,s,bits"
      long p, sc, bits = 128;
      const size_t c1 = tail.find(',');
      if (c1 == std::string::npos) return 1;
      const size_t c2 = tail.find(',', c1 + 1);
      const char* b = tail.c_str();
      if (!parse_int(b, b + c1, &p)) return 1;
      if (!parse_int(b + c1 + 1, c2 == std::string::npos ? b + tail.size() : b + c2, &sc)) return 1;
      if (c2 != std::string::npos && !parse_int(b + c2 + 1, b + tail.size(), &bits)) return 1;
      switch (bits) {
        case 32: *out = AH_INT32; return 0;
        case 64: *out = AH_INT64; return 0;
        case 128: *out = AH_FIXED16; return 0;
        case 256: *out = AH_FIXED32; return 0;
        default: *why = "Only 32/64/128/256 bit wide decimals are supported in the Rust implementation"; return 3;
      }
    }
    if (head == "w") {  // FixedSizeBinary(n): bit-copy kernels exist for the native widths
      long n;
      if (!parse_int(tail.c_str(), tail.c_str() + tail.size(), &n)) return 1;
      switch (n) {
        case 1: *out = AH_UINT8; return 0;
        case 2: *out = AH_UINT16; return 0;
        case 4: *out = AH_UINT32; return 0;
        case 8: *out = AH_UINT64; return 0;
        case 16: *out = AH_FIXED16; return 0;
        case 32: *out = AH_FIXED32; return 0;
        default: *why = "FixedSizeBinary(" + std::to_string(n) + ") has no device kernels"; return 2;
      }
    }
  }
  return 1;
}

ah_status type_from_format(ah_context* ctx, const char* format, ah_type* out) {
  if (!format) return ah_fail(ctx, AH_C_DATA_INTERFACE, "Null pointer passed where a format string was expected");
  std::string why;
  switch (format_to_type(format, out, &why)) {
    case 0: return AH_OK;
    case 2: return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "%s", why.c_str());
    case 3: return ah_fail(ctx, AH_C_DATA_INTERFACE, "%s", why.c_str());
    default:  // ffi.rs:689-691 — `{other:?}` is the Debug form of a &str, i.e. quoted again
      return ah_fail(ctx, AH_C_DATA_INTERFACE,
                     "The datatype \"\"%s\"\" is still not supported in Rust implementation", format);
  }
}

struct HostArrayPrivate {
  const void* buffers[3] = {nullptr, nullptr, nullptr};
  void* owned[3] = {nullptr, nullptr, nullptr};
};
void release_host_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  auto* p = static_cast<HostArrayPrivate*>(a->private_data);
  if (p) {
    for (void* o : p->owned) free(o);
    delete p;
  }
  a->release = nullptr;  // arrow-data/src/ffi.rs:96
  a->private_data = nullptr;
}
void release_host_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  free(const_cast<char*>(s->format));
  s->format = nullptr;
  s->release = nullptr;
}

// D2H of `bytes` into a fresh host buffer (64-byte aligned like the reference's allocations,
// arrow-buffer/src/alloc/alignment.rs).  Zero-length buffers still get a valid pointer.
ah_status host_copy(ah_context* ctx, const void* dev, size_t bytes, size_t pad_to, void** out) {
  const size_t cap = ((std::max(bytes, pad_to) + 63) / 64 + 1) * 64;
  void* h = aligned_alloc(64, cap);
  if (!h) return ah_fail(ctx, AH_OUT_OF_MEMORY, "host allocation of %zu bytes failed", cap);
  memset(static_cast<char*>(h) + bytes, 0, cap - bytes);
  if (bytes) {
    ctx->stats.device_to_host_bytes += (int64_t)bytes;
    hipError_t e = hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) {
      free(h);
      return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in C Data export", hipGetErrorString(e));
    }
  }
  *out = h;
  return AH_OK;
}

// bits [bit_offset, bit_offset+len) of a DEVICE bitmap -> fresh host bytes starting at bit 0
ah_status host_copy_bits(ah_context* ctx, const uint8_t* dev, int64_t bit_offset, int64_t len, void** out) {
  const size_t nbytes = (size_t)ah_ceil_div(len, 8);
  if ((bit_offset & 7) == 0) return host_copy(ctx, dev + bit_offset / 8, nbytes, 0, out);
  void* tmp = nullptr;  // align_nulls (arrow-data/src/ffi.rs:104-121) as a device funnel shift
  const size_t tb = ah_bitmap_bytes(len);
  AH_TRY(ah_pool_alloc(ctx, tb, &tmp));
  ah_status st = ah_bitmap_op(ctx, BM_COPY, make_bitview(dev, bit_offset), BitView{nullptr, 0}, BitView{nullptr, 0},
                              len, (unsigned long long*)tmp, nullptr);
  if (st == AH_OK) st = host_copy(ctx, tmp, nbytes, 0, out);
  ah_pool_free(ctx, tmp);
  return st;
}

}  // namespace

extern "C" ah_status ah_type_from_format(ah_context* ctx, const char* format, ah_type* out) {
  ah_ctx_guard _guard(ctx);
  if (!out) return AH_INVALID_ARGUMENT;  // ctx may be NULL: status only, no message
  return type_from_format(ctx, format, out);
}

extern "C" const char* ah_format_of_type(ah_type t) {  // arrow-schema/src/ffi.rs:779-856
  switch (t) {
    case AH_BOOL: return "b";
    case AH_INT8: return "c"; case AH_UINT8: return "C";
    case AH_INT16: return "s"; case AH_UINT16: return "S";
    case AH_INT32: return "i"; case AH_UINT32: return "I";
    case AH_INT64: return "l"; case AH_UINT64: return "L";
    case AH_FLOAT16: return "e"; case AH_FLOAT32: return "f"; case AH_FLOAT64: return "g";
    case AH_FIXED16: return "d:38,0";
    case AH_FIXED32: return "d:76,0,256";
    case AH_UTF8: return "u"; case AH_LARGE_UTF8: return "U";
    default: return nullptr;
  }
}

extern "C" ah_status ah_import_c_data(ah_context* ctx, const struct ArrowArray* array,
                                      const struct ArrowSchema* schema, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !array || !schema || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (!array->release) return ah_fail(ctx, AH_C_DATA_INTERFACE, "The ArrowArray has already been released");
  ah_type t;
  AH_TRY(type_from_format(ctx, schema->format, &t));
  if (schema->dictionary || array->dictionary)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "dictionary-encoded C Data arrays have no device kernels");
  const bool is_str = (t == AH_UTF8 || t == AH_LARGE_UTF8);
  const int64_t want = is_str ? 3 : 2;
  const int64_t len = array->length, off = array->offset;
  if (len < 0 || off < 0) return ah_fail(ctx, AH_C_DATA_INTERFACE, "negative length or offset");
  if (array->n_buffers != want)  // bit_width(), arrow-array/src/ffi.rs:119-214
    return ah_fail(ctx, AH_C_DATA_INTERFACE,
                   "The datatype \"%s\" expects %lld buffers, but requested %lld. Please verify that the C data "
                   "interface is correctly implemented.",
                   ah_type_name(t), (long long)want, (long long)array->n_buffers);
  const void* const* bufs = array->buffers;
  out->type = t;
  out->length = len;

  auto fail = [&](ah_status st) {
    ah_array_release(ctx, out);
    return st;
  };
  auto h2d = [&](const void* src, size_t bytes, size_t alloc_bytes, void** dst) -> ah_status {
    AH_TRY(ah_out_alloc(ctx, alloc_bytes, dst));
    if (bytes) {
      ctx->stats.host_to_device_bytes += (int64_t)bytes;
      hipError_t e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in C Data import", hipGetErrorString(e));
    }
    return AH_OK;
  };

  // validity: the bytes covering bits [off, off+len); the sub-byte remainder stays as a bit offset
  const uint8_t* hv = static_cast<const uint8_t*>(bufs[0]);
  if (hv && len > 0) {
    const size_t nb = (size_t)ah_ceil_div((off & 7) + len, 8);
    const size_t cap = ah_bitmap_bytes((off & 7) + len) + 8;
    void* dv = nullptr;
    ah_status st = h2d(hv + off / 8, nb, cap, &dv);
    out->validity = static_cast<uint8_t*>(dv);
    out->validity_bytes = (int64_t)cap;
    out->validity_bit_offset = off & 7;
    if (st != AH_OK) return fail(st);
  }

  if (t == AH_BOOL) {
    const uint8_t* hb = static_cast<const uint8_t*>(bufs[1]);
    if (!hb && len > 0) return fail(ah_fail(ctx, AH_C_DATA_INTERFACE, "The external buffer at position 1 is null."));
    if (len > 0) {
      const size_t nb = (size_t)ah_ceil_div((off & 7) + len, 8);
      const size_t cap = ah_bitmap_bytes((off & 7) + len) + 8;
      ah_status st = h2d(hb + off / 8, nb, cap, &out->values);
      out->values_bytes = (int64_t)cap;
      out->values_bit_offset = off & 7;
      if (st != AH_OK) return fail(st);
    }
  } else if (is_str) {
    const int ow = t == AH_UTF8 ? 4 : 8;
    const uint8_t* ho = static_cast<const uint8_t*>(bufs[1]);
    const uint8_t* hd = static_cast<const uint8_t*>(bufs[2]);
    if (!ho && len > 0) return fail(ah_fail(ctx, AH_C_DATA_INTERFACE, "The external buffer at position 1 is null."));
    if (ho) {
      // buffer_len (ffi.rs:470-500) sizes the data buffer by the last offset; only the bytes the
      // imported rows reference travel, with the offsets rebased to start at 0.
      ho += (size_t)off * ow;
      int64_t first, last;
      if (ow == 4) {
        first = reinterpret_cast<const int32_t*>(ho)[0];
        last = reinterpret_cast<const int32_t*>(ho)[len];
      } else {
        first = reinterpret_cast<const int64_t*>(ho)[0];
        last = reinterpret_cast<const int64_t*>(ho)[len];
      }
      if (first < 0 || last < first) return fail(ah_fail(ctx, AH_C_DATA_INTERFACE, "offsets are not monotonic"));
      if (!hd && last > first)
        return fail(ah_fail(ctx, AH_C_DATA_INTERFACE, "The external buffer at position 2 is null."));
      const size_t ob = (size_t)(len + 1) * ow;
      std::vector<uint8_t> rebased;  // a sliced producer array: ship only the referenced bytes
      if (first != 0) {
        rebased.resize(ob);
        if (ow == 4) {
          const int32_t* src = reinterpret_cast<const int32_t*>(ho);
          int32_t* dst = reinterpret_cast<int32_t*>(rebased.data());
          for (int64_t i = 0; i <= len; ++i) dst[i] = src[i] - (int32_t)first;
        } else {
          const int64_t* src = reinterpret_cast<const int64_t*>(ho);
          int64_t* dst = reinterpret_cast<int64_t*>(rebased.data());
          for (int64_t i = 0; i <= len; ++i) dst[i] = src[i] - first;
        }
      }
      ah_status st = h2d(first ? rebased.data() : ho, ob, ob, &out->offsets);
      out->offsets_bytes = (int64_t)ob;
      if (st != AH_OK) return fail(st);
      if (first) {  // the staging vector dies at the end of this scope
        hipError_t e = ah_stream_wait(ctx);
        if (e != hipSuccess)
          return fail(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in C Data import", hipGetErrorString(e)));
      }
      const size_t db = (size_t)(last - first);
      void* dd = nullptr;
      st = ah_out_alloc(ctx, db ? db : 8, &dd);
      out->values = dd;
      out->values_bytes = (int64_t)(db ? db : 8);
      if (st != AH_OK) return fail(st);
      if (db) {
        ctx->stats.host_to_device_bytes += (int64_t)db;
        hipError_t e = hipMemcpyAsync(dd, hd + first, db, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess)
          return fail(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in C Data import", hipGetErrorString(e)));
      }
    }
  } else {
    const int w = ah_type_width(t);
    const uint8_t* hvals = static_cast<const uint8_t*>(bufs[1]);
    if (!hvals && len > 0) return fail(ah_fail(ctx, AH_C_DATA_INTERFACE, "The external buffer at position 1 is null."));
    if (len > 0) {
      const size_t vb = (size_t)len * w;
      ah_status st = h2d(hvals + (size_t)off * w, vb, vb, &out->values);
      out->values_bytes = (int64_t)vb;
      if (st != AH_OK) return fail(st);
    }
  }
  hipError_t e = ah_stream_wait(ctx);
  if (e != hipSuccess) return fail(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in C Data import", hipGetErrorString(e)));

  if (!out->validity) {
    out->null_count = 0;
  } else if (array->null_count >= 0) {
    out->null_count = array->null_count;
  } else {
    int64_t set = 0;
    ah_status st = ah_count_set_bits(ctx, out->validity, out->validity_bit_offset, len, &set);
    if (st != AH_OK) return fail(st);
    out->null_count = len - set;
  }
  return AH_OK;
}

extern "C" ah_status ah_export_c_data(ah_context* ctx, const ah_array_view* v, const char* format,
                                      struct ArrowArray* oa, struct ArrowSchema* os) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !oa || !os) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  const ah_type t = v->type;
  if (format) {
    ah_type ft;
    AH_TRY(type_from_format(ctx, format, &ft));
    const bool same = ft == t || (ah_type_width(ft) > 0 && ah_type_width(ft) == ah_type_width(t));
    if (!same)
      return ah_fail(ctx, AH_C_DATA_INTERFACE, "format \"%s\" does not describe the physical layout %s", format,
                     ah_type_name(t));
  } else if (!(format = ah_format_of_type(t))) {
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown array type %d", (int)t);
  }
  const int64_t len = v->length;
  const bool is_str = (t == AH_UTF8 || t == AH_LARGE_UTF8);
  auto* priv = new HostArrayPrivate();
  auto fail = [&](ah_status st) {
    for (void* o : priv->owned) free(o);
    delete priv;
    return st;
  };

  int64_t nulls = 0;
  if (v->validity) {
    nulls = v->null_count;
    if (nulls < 0) {
      int64_t set = 0;
      ah_status st = ah_count_set_bits(ctx, v->validity, v->validity_bit_offset, len, &set);
      if (st != AH_OK) return fail(st);
      nulls = len - set;
    }
    ah_status st = host_copy_bits(ctx, v->validity, v->validity_bit_offset, len, &priv->owned[0]);
    if (st != AH_OK) return fail(st);
  }
  ah_status st = AH_OK;
  if (t == AH_BOOL) {
    st = host_copy_bits(ctx, static_cast<const uint8_t*>(v->values), v->values_bit_offset, len, &priv->owned[1]);
  } else if (is_str) {
    const int ow = t == AH_UTF8 ? 4 : 8;
    st = host_copy(ctx, v->offsets, v->offsets ? (size_t)(len + 1) * ow : 0, ow, &priv->owned[1]);
    if (st == AH_OK) {
      int64_t last = 0;  // an absent offsets buffer (empty array) exports the single offset 0
      if (v->offsets)
        last = ow == 4 ? (int64_t)static_cast<int32_t*>(priv->owned[1])[len] : static_cast<int64_t*>(priv->owned[1])[len];
      st = host_copy(ctx, v->values, (size_t)last, 0, &priv->owned[2]);
    }
  } else {
    const int w = ah_type_width(t);
    if (w <= 0) return fail(ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown array type %d", (int)t));
    st = host_copy(ctx, v->values, (size_t)len * w, 0, &priv->owned[1]);
  }
  if (st != AH_OK) return fail(st);
  for (int i = 0; i < 3; ++i) priv->buffers[i] = priv->owned[i];

  char* fmt = strdup(format);
  if (!fmt) return fail(ah_fail(ctx, AH_OUT_OF_MEMORY, "host allocation failed"));
  oa->length = len;
  oa->null_count = nulls;
  oa->offset = 0;
  oa->n_buffers = is_str ? 3 : 2;
  oa->n_children = 0;
  oa->buffers = priv->buffers;
  oa->children = nullptr;
  oa->dictionary = nullptr;
  oa->release = release_host_array;
  oa->private_data = priv;
  os->format = fmt;
  os->name = "";
  os->metadata = nullptr;
  os->flags = ARROW_FLAG_NULLABLE;
  os->n_children = 0;
  os->children = nullptr;
  os->dictionary = nullptr;
  os->release = release_host_schema;
  os->private_data = nullptr;
  return AH_OK;
}

// ------------------------------------------------------------------ Arrow C Device Data Interface
namespace {

struct DeviceArrayPrivate {
  const void* buffers[3] = {nullptr, nullptr, nullptr};
  ah_context* ctx = nullptr;
  ah_array_out owned{};       // buffers that moved in (flags / pointers as the kernel returned them)
  void* realigned[2] = {nullptr, nullptr};  // re-aligned validity / Boolean bitmaps (pool allocations)
};
void release_device_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  auto* p = static_cast<DeviceArrayPrivate*>(a->private_data);
  if (p) {
    if (p->ctx) {
      hipSetDevice(p->ctx->device);
      ah_array_release(p->ctx, &p->owned);
      for (void* r : p->realigned)
        if (r) ah_pool_free(p->ctx, r);
    }
    delete p;
  }
  a->release = nullptr;
  a->private_data = nullptr;
}

// device bitmap [bit_offset, bit_offset+len) as a pointer to a bit-0-aligned bitmap (re-aligned copy if needed)
ah_status aligned_bits(ah_context* ctx, const void* bits, int64_t bit_offset, int64_t len, const void** out, void** owned) {
  *owned = nullptr;
  if ((bit_offset & 7) == 0) {
    *out = static_cast<const uint8_t*>(bits) + bit_offset / 8;
    return AH_OK;
  }
  void* tmp = nullptr;
  AH_TRY(ah_pool_alloc(ctx, ah_bitmap_bytes(len), &tmp));
  ah_status st = ah_bitmap_op(ctx, BM_COPY, make_bitview(bits, bit_offset), BitView{nullptr, 0}, BitView{nullptr, 0}, len,
                              (unsigned long long*)tmp, nullptr);
  if (st == AH_OK && ah_stream_wait(ctx) != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "bitmap re-alignment failed");
  if (st != AH_OK) {
    ah_pool_free(ctx, tmp);
    return st;
  }
  *out = tmp;
  *owned = tmp;
  return AH_OK;
}

}  // namespace

extern "C" ah_status ah_export_c_device_data(ah_context* ctx, const ah_array_view* v, ah_array_out* owned,
                                             const char* format, struct ArrowDeviceArray* od, struct ArrowSchema* os) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !od || !os) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  const ah_type t = v->type;
  if (format) {
    ah_type ft;
    AH_TRY(type_from_format(ctx, format, &ft));
    const bool same = ft == t || (ah_type_width(ft) > 0 && ah_type_width(ft) == ah_type_width(t));
    if (!same) return ah_fail(ctx, AH_C_DATA_INTERFACE, "format \"%s\" does not describe the physical layout %s", format, ah_type_name(t));
  } else if (!(format = ah_format_of_type(t))) {
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown array type %d", (int)t);
  }
  if (t == AH_UTF8_VIEW || t == AH_BINARY_VIEW)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "device export of %s (variadic buffers stay with the host)", ah_type_name(t));
  const int64_t len = v->length;
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;
  auto* priv = new DeviceArrayPrivate();
  priv->ctx = ctx;
  auto fail = [&](ah_status st) {
    for (void* r : priv->realigned)
      if (r) ah_pool_free(ctx, r);
    delete priv;
    return st;
  };
  int64_t nulls = 0;
  ah_status st = ah_resolve_null_count(ctx, v, &nulls);
  if (st != AH_OK) return fail(st);
  if (v->validity && len > 0) {
    st = aligned_bits(ctx, v->validity, v->validity_bit_offset, len, &priv->buffers[0], &priv->realigned[0]);
    if (st != AH_OK) return fail(st);
  }
  if (t == AH_BOOL) {
    if (len > 0) {
      st = aligned_bits(ctx, v->values, v->values_bit_offset, len, &priv->buffers[1], &priv->realigned[1]);
      if (st != AH_OK) return fail(st);
    }
  } else if (is_str) {
    priv->buffers[1] = v->offsets;
    priv->buffers[2] = v->values;
  } else {
    priv->buffers[1] = v->values;
  }
  char* fmt = strdup(format);
  if (!fmt) return fail(ah_fail(ctx, AH_OUT_OF_MEMORY, "host allocation failed"));
  if (owned) {  // the buffers move into the exported struct
    priv->owned = *owned;
    ah_out_init(owned);
  } else {
    ah_out_init(&priv->owned);
  }
  memset(od, 0, sizeof *od);
  od->array.length = len;
  od->array.null_count = v->validity ? nulls : 0;
  od->array.offset = 0;
  od->array.n_buffers = is_str ? 3 : 2;
  od->array.buffers = priv->buffers;
  od->array.release = release_device_array;
  od->array.private_data = priv;
  od->device_id = ctx->device;
  od->device_type = ARROW_DEVICE_ROCM;
  od->sync_event = nullptr;
  os->format = fmt;
  os->name = "";
  os->metadata = nullptr;
  os->flags = ARROW_FLAG_NULLABLE;
  os->n_children = 0;
  os->children = nullptr;
  os->dictionary = nullptr;
  os->release = release_host_schema;
  os->private_data = nullptr;
  return AH_OK;
}

extern "C" ah_status ah_import_c_device_data(ah_context* ctx, const struct ArrowDeviceArray* d, const struct ArrowSchema* schema,
                                             ah_array_view* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !d || !schema || !out) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  memset(out, 0, sizeof *out);
  const struct ArrowArray* a = &d->array;
  if (!a->release) return ah_fail(ctx, AH_C_DATA_INTERFACE, "The ArrowArray has already been released");
  if (d->device_type != ARROW_DEVICE_ROCM && d->device_type != ARROW_DEVICE_ROCM_HOST)
    return ah_fail(ctx, AH_C_DATA_INTERFACE, "device type %d is not ROCm memory", (int)d->device_type);
  if (d->device_type == ARROW_DEVICE_ROCM && d->device_id != ctx->device)
    return ah_fail(ctx, AH_C_DATA_INTERFACE, "array lives on device %lld, the context on device %d", (long long)d->device_id, ctx->device);
  ah_type t;
  AH_TRY(type_from_format(ctx, schema->format, &t));
  if (schema->dictionary || a->dictionary)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "dictionary-encoded C Data arrays have no device kernels");
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;
  const int64_t want = is_str ? 3 : 2;
  if (a->n_buffers != want)
    return ah_fail(ctx, AH_C_DATA_INTERFACE,
                   "The datatype \"%s\" expects %lld buffers, but requested %lld. Please verify that the C data "
                   "interface is correctly implemented.", ah_type_name(t), (long long)want, (long long)a->n_buffers);
  if (a->length < 0 || a->offset < 0) return ah_fail(ctx, AH_C_DATA_INTERFACE, "negative length or offset");
  if (d->sync_event)  // the producer's work must be visible to kernels this context launches
    AH_HIP(ctx, hipStreamWaitEvent(ctx->stream, *static_cast<hipEvent_t*>(d->sync_event), 0));
  const int64_t off = a->offset, len = a->length;
  out->type = t;
  out->length = len;
  out->validity = static_cast<const uint8_t*>(a->buffers[0]);
  out->validity_bit_offset = out->validity ? off : 0;
  out->null_count = out->validity ? a->null_count : 0;  // -1 stays "unknown": counted on first use
  if (len > 0 && !a->buffers[1]) return ah_fail(ctx, AH_C_DATA_INTERFACE, "The external buffer at position 1 is null.");
  if (t == AH_BOOL) {
    out->values = a->buffers[1];
    out->values_bit_offset = off;
  } else if (is_str) {
    const int ow = t == AH_UTF8 ? 4 : 8;
    out->offsets = a->buffers[1] ? static_cast<const uint8_t*>(a->buffers[1]) + off * ow : nullptr;
    out->values = a->buffers[2];
  } else {
    out->values = a->buffers[1] ? static_cast<const uint8_t*>(a->buffers[1]) + off * ah_type_width(t) : nullptr;
  }
  return AH_OK;
}
