// concat.hip — arrow_select::concat for primitive / boolean arrays on MI355X.
//
// Reference: concat (arrow-select/src/concat.rs:495) -> concat_primitives
// :334-343 (values appended piece by piece; NullBufferBuilder appends each
// piece's validity, materialising a buffer only if some piece has nulls).
// Also the reassembly primitive of the row-sharded multi-GPU path: value pieces
// are device-to-device copies straight to their final offset; validity pieces
// land at bit offset sum(len_<r) — generally not byte aligned — through the
// funnel-shift merge kernel in bitmap.hip (reference analogue
// arrow-buffer/src/util/bit_mask.rs:33 set_bits).
#include "common.hpp"

#include <algorithm>

#include <vector>

namespace {

// GenericByteBuilder::append_array (arrow-array/src/builder/generic_bytes_builder.rs:169-206): offsets shifted so
// the piece continues at the builder's next offset
template <typename O>
__global__ void shift_offsets_kernel(const O* in, O* out, int64_t n, O first, O base) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (O)(in[i] - first + base);
}

// concat_bytes (arrow-select/src/concat.rs:355-368)
ah_status concat_strings(ah_context* ctx, int32_t n, const ah_array_view* pieces, bool any_nulls, int64_t total,
                         ah_array_out* out) {
  const ah_type t = pieces[0].type;
  const int ow = t == AH_UTF8 ? 4 : 8;
  std::vector<int64_t> first(n, 0), last(n, 0);
  for (int i0 = 0; i0 < n; i0 += 64) {  // 2 scalars per piece through the pinned slots, 64 pieces per round trip
    const int cnt = std::min(64, n - i0);
    for (int j = 0; j < cnt; ++j) {
      const ah_array_view& p = pieces[i0 + j];
      ctx->pinned[2 * j] = ctx->pinned[2 * j + 1] = 0;
      if (p.length == 0) continue;
      AH_HIP(ctx, ah_d2h(ctx, ctx->pinned + 2 * j, p.offsets, ow));
      AH_HIP(ctx, ah_d2h(ctx, ctx->pinned + 2 * j + 1, (const char*)p.offsets + p.length * ow, ow));
    }
    AH_HIP(ctx, ah_stream_wait(ctx));
    for (int j = 0; j < cnt; ++j) {
      first[i0 + j] = ow == 4 ? (int64_t)(int32_t)ctx->pinned[2 * j] : (int64_t)ctx->pinned[2 * j];
      last[i0 + j] = ow == 4 ? (int64_t)(int32_t)ctx->pinned[2 * j + 1] : (int64_t)ctx->pinned[2 * j + 1];
    }
  }
  int64_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    bytes += last[i] - first[i];
    if (ow == 4 && bytes > INT32_MAX)  // :185-189 `OffsetOverflowError(shift + last offset)`
      return ah_fail(ctx, AH_OFFSET_OVERFLOW_ERROR, "%lld", (long long)bytes);
  }
  const size_t obytes = (size_t)(total + 1) * ow, dbytes = (size_t)std::max<int64_t>(bytes, 8);
  const size_t bbytes = any_nulls ? ah_bitmap_bytes(total) : 0;
  void *oo = nullptr, *od = nullptr, *ob = nullptr;
  ah_status st = ah_out_alloc(ctx, obytes, &oo);
  if (st == AH_OK) st = ah_out_alloc(ctx, dbytes, &od);
  if (st == AH_OK && any_nulls) st = ah_out_alloc(ctx, bbytes, &ob);
  auto cleanup = [&](ah_status s) {
    ah_out_free(ctx, oo, obytes);
    ah_out_free(ctx, od, dbytes);
    ah_out_free(ctx, ob, bbytes);
    return s;
  };
  if (st != AH_OK) return cleanup(st);
  if (any_nulls) hipMemsetAsync(ob, 0, bbytes, ctx->stream);
  int64_t pos = 0, base = 0, valid_total = 0;
  ah_prof_scope ps(ctx, "concat");
  for (int i = 0; i < n && st == AH_OK; ++i) {
    const ah_array_view& p = pieces[i];
    if (p.length == 0) continue;
    const int64_t cnt = p.length + 1;
    const unsigned grid = (unsigned)((cnt + 255) / 256);
    if (ow == 4)
      hipLaunchKernelGGL(shift_offsets_kernel<int32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int32_t*)p.offsets,
                         (int32_t*)oo + pos, cnt, (int32_t)first[i], (int32_t)base);
    else
      hipLaunchKernelGGL(shift_offsets_kernel<int64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t*)p.offsets,
                         (int64_t*)oo + pos, cnt, first[i], base);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && last[i] > first[i])
      e = hipMemcpyAsync((char*)od + base, (const char*)p.values + first[i], (size_t)(last[i] - first[i]),
                         hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "concat copy failed: %s", hipGetErrorString(e));
    if (st == AH_OK && any_nulls) {
      int64_t set = 0;
      st = ah_bitmap_set_bits(ctx, (uint8_t*)ob, pos, p.validity, p.validity_bit_offset, p.length, &set);
      valid_total += set;
    }
    pos += p.length;
    base += last[i] - first[i];
  }
  hipError_t e = ah_stream_wait(ctx);
  if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "concat failed: %s", hipGetErrorString(e));
  if (st != AH_OK) return cleanup(st);
  out->offsets = oo;
  out->offsets_bytes = (int64_t)obytes;
  out->values = od;
  out->values_bytes = (int64_t)dbytes;
  if (any_nulls) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = total - valid_total;
  }
  return AH_OK;
}

}  // namespace

extern "C" ah_status ah_concat(ah_context* ctx, int32_t n, const ah_array_view* pieces,
                               ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out || (n > 0 && !pieces)) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (n <= 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "concat requires input of at least one array");
  const ah_type t = pieces[0].type;
  const int w = ah_type_width(t);
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;
  if ((w < 0 && !is_str) || t == AH_UTF8_VIEW || t == AH_BINARY_VIEW)  // views: buffer indices would need renumbering
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "concat not supported for type %s", ah_type_name(t));
  int64_t total = 0;
  bool any_nulls = false;
  for (int i = 1; i < n; ++i) {
    if (pieces[i].type == t) continue;
    // concat.rs:505-535: up to 10 unique data types in order of appearance, ", ..." once an 11th shows up
    std::string msg = std::string("It is not possible to concatenate arrays of different data types (") + ah_type_name(t);
    std::vector<int32_t> seen{(int32_t)t};
    for (int j = 0; j < n; ++j) {
      const bool unique = std::find(seen.begin(), seen.end(), (int32_t)pieces[j].type) == seen.end();
      if (unique) seen.push_back((int32_t)pieces[j].type);
      if (seen.size() == 11) {
        msg += ", ...";
        break;
      }
      if (unique) msg += std::string(", ") + ah_type_name(pieces[j].type);
    }
    msg += ").";
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "%s", msg.c_str());
  }
  for (int i = 0; i < n; ++i) {
    total += pieces[i].length;
    int64_t nulls = 0;
    AH_TRY(ah_resolve_null_count(ctx, &pieces[i], &nulls));
    if (pieces[i].validity && nulls > 0) any_nulls = true;
  }
  out->type = t;
  out->length = total;
  if (total == 0) return AH_OK;
  if (is_str) {
    ah_status st = concat_strings(ctx, n, pieces, any_nulls, total, out);
    if (st != AH_OK) ah_out_init(out);
    return st;
  }
  size_t vbytes = w ? (size_t)total * w : ah_bitmap_bytes(total);
  size_t bbytes = any_nulls ? ah_bitmap_bytes(total) : 0;
  void* ov = nullptr;
  void* ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  if (any_nulls) {
    ah_status st = ah_out_alloc(ctx, bbytes, &ob);
    if (st != AH_OK) {
      ah_out_free(ctx, ov, vbytes);
      return st;
    }
    hipMemsetAsync(ob, 0, bbytes, ctx->stream);
  }
  if (w == 0) hipMemsetAsync(ov, 0, vbytes, ctx->stream);
  int64_t pos = 0, valid_total = 0;
  ah_status st = AH_OK;
  ah_prof_scope ps(ctx, "concat");
  for (int i = 0; i < n && st == AH_OK; ++i) {
    const ah_array_view* p = &pieces[i];
    if (p->length == 0) continue;
    if (w) {
      hipError_t e = hipMemcpyAsync((char*)ov + (size_t)pos * w, p->values, (size_t)p->length * w,
                                    hipMemcpyDeviceToDevice, ctx->stream);
      if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "concat copy failed: %s", hipGetErrorString(e));
    } else {
      st = ah_bitmap_set_bits(ctx, (uint8_t*)ov, pos, (const uint8_t*)p->values, p->values_bit_offset,
                              p->length, nullptr);
    }
    if (st == AH_OK && any_nulls) {
      int64_t set = 0;
      st = ah_bitmap_set_bits(ctx, (uint8_t*)ob, pos, p->validity, p->validity_bit_offset, p->length, &set);
      valid_total += set;
    }
    pos += p->length;
  }
  hipError_t e = ah_stream_wait(ctx);
  if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "concat failed: %s", hipGetErrorString(e));
  if (st != AH_OK) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    return st;
  }
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (any_nulls) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = total - valid_total;
  }
  return AH_OK;
}

// InProgressPrimitiveArray::copy_rows (arrow-select/src/coalesce/primitive.rs): append rows
// [offset, offset+len) of `src` to an in-progress builder at row `dst_row_offset`.
extern "C" ah_status ah_copy_rows_into(ah_context* ctx, const ah_array_view* src, int64_t offset, int64_t len,
                                       void* dst_values, uint8_t* dst_validity, int64_t dst_row_offset,
                                       int64_t* appended_nulls) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !src || !dst_values || !dst_validity) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  if (appended_nulls) *appended_nulls = 0;
  const int w = ah_type_width(src->type);
  if (w <= 0)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "copy_rows not supported for type %s", ah_type_name(src->type));
  if (offset < 0 || len < 0 || offset + len > src->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "copy_rows range [%lld, %lld) exceeds source length %lld",
                   (long long)offset, (long long)(offset + len), (long long)src->length);
  if (len == 0) return AH_OK;
  ah_prof_scope ps(ctx, "copy_rows");
  AH_HIP(ctx, hipMemcpyAsync((char*)dst_values + (size_t)dst_row_offset * w,
                             (const char*)src->values + (size_t)offset * w, (size_t)len * w,
                             hipMemcpyDeviceToDevice, ctx->stream));
  int64_t set = len;
  // a source without a null buffer appends `len` valid rows (NullBufferBuilder::append_n_non_nulls)
  AH_TRY(ah_bitmap_set_bits(ctx, dst_validity, dst_row_offset, src->validity,
                            src->validity ? src->validity_bit_offset + offset : 0, len,
                            src->validity ? &set : nullptr));
  if (!src->validity) AH_HIP(ctx, ah_stream_wait(ctx));
  if (appended_nulls) *appended_nulls = len - set;
  return AH_OK;
}

// copy_rows without a host wait: *nulls_acc (device) += the number of null rows appended
extern "C" ah_status ah_copy_rows_into_acc(ah_context* ctx, const ah_array_view* src, int64_t offset, int64_t len,
                                           void* dst_values, uint8_t* dst_validity, int64_t dst_row_offset,
                                           uint64_t* nulls_acc) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !src || !dst_values || !dst_validity || !nulls_acc) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  const int w = ah_type_width(src->type);
  if (w <= 0)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "copy_rows not supported for type %s", ah_type_name(src->type));
  if (offset < 0 || len < 0 || offset + len > src->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "copy_rows range [%lld, %lld) exceeds source length %lld",
                   (long long)offset, (long long)(offset + len), (long long)src->length);
  if (len == 0) return AH_OK;
  ctx->inflight = true;
  ah_prof_scope ps(ctx, "copy_rows");
  AH_HIP(ctx, hipMemcpyAsync((char*)dst_values + (size_t)dst_row_offset * w,
                             (const char*)src->values + (size_t)offset * w, (size_t)len * w,
                             hipMemcpyDeviceToDevice, ctx->stream));
  return ah_bitmap_set_bits_acc(ctx, dst_validity, dst_row_offset, src->validity,
                                src->validity ? src->validity_bit_offset + offset : 0, len, (unsigned long long*)nulls_acc);
}

// `n` (<= 200) device words -> host in ONE wait, optionally resetting them to zero: the read side of the *_acc calls
extern "C" ah_status ah_read_words(ah_context* ctx, uint64_t* dev_words, int32_t n, uint64_t* host_out, int32_t reset) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || n < 0 || n > 200 || (n > 0 && (!dev_words || !host_out))) return AH_INVALID_ARGUMENT;
  if (n == 0) return AH_OK;
  hipSetDevice(ctx->device);
  AH_HIP(ctx, ah_d2h_wait(ctx, ctx->pinned + 16, dev_words, (size_t)n * 8, reset != 0, 0));
  memcpy(host_out, ctx->pinned + 16, (size_t)n * 8);
  return AH_OK;
}

// arrow_select::window::shift (arrow-select/src/window.rs:56-80): offset 0 is a clone (zero copy), |offset| >= len
// (or i64::MIN) is new_null_array(len), otherwise concat(nulls(k), slice(0, len - k)) for a right shift or
// concat(slice(k, len - k), nulls(k)) for a left shift.  Built exactly that way: the null piece is a zeroed
// scratch array, the data piece a sub-view, and ah_concat does the copies and the bit-shifted validity merge.
extern "C" ah_status ah_shift(ah_context* ctx, const ah_array_view* values, int64_t offset, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  const ah_type t = values->type;
  const int w = ah_type_width(t);
  const bool is_str = t == AH_UTF8 || t == AH_LARGE_UTF8;
  if ((w < 0 && !is_str) || t == AH_UTF8_VIEW || t == AH_BINARY_VIEW)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "shift not supported for type %s", ah_type_name(t));
  const int64_t len = values->length;
  out->type = t;
  out->length = len;
  if (offset == 0) {  // make_array(array.to_data()): the same buffers
    out->values = const_cast<void*>(values->values);
    out->values_bit_offset = values->values_bit_offset;
    out->values_bytes = w > 0 ? len * w : 0;
    out->offsets = const_cast<void*>(values->offsets);
    out->flags = AH_OUT_BORROWED;
    if (values->validity) {
      int64_t nulls = 0;
      AH_TRY(ah_resolve_null_count(ctx, values, &nulls));
      out->validity = const_cast<uint8_t*>(values->validity);
      out->validity_bit_offset = values->validity_bit_offset;
      out->null_count = nulls;
    }
    return AH_OK;
  }
  if (len == 0) return AH_OK;
  const size_t ow = t == AH_UTF8 ? 4 : 8;
  const bool all_null = offset == INT64_MIN || (offset < 0 ? -offset : offset) >= len;
  if (all_null) {  // new_null_array: zeroed values / offsets, every validity bit clear
    const size_t vbytes = is_str ? 0 : (w ? (size_t)len * w : ah_bitmap_bytes(len));
    const size_t bbytes = ah_bitmap_bytes(len), obytes = is_str ? (size_t)(len + 1) * ow : 0;
    void *ov = nullptr, *ob = nullptr, *oo = nullptr;
    ah_status st = ah_out_alloc(ctx, std::max<size_t>(vbytes, 8), &ov);
    if (st == AH_OK) st = ah_out_alloc(ctx, bbytes, &ob);
    if (st == AH_OK && is_str) st = ah_out_alloc(ctx, obytes, &oo);
    hipError_t e = hipSuccess;
    if (st == AH_OK) e = hipMemsetAsync(ov, 0, std::max<size_t>(vbytes, 8), ctx->stream);
    if (st == AH_OK && e == hipSuccess) e = hipMemsetAsync(ob, 0, bbytes, ctx->stream);
    if (st == AH_OK && e == hipSuccess && oo) e = hipMemsetAsync(oo, 0, obytes, ctx->stream);
    if (st == AH_OK && e == hipSuccess) e = ah_end_of_call_sync(ctx);
    if (st == AH_OK && e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in shift", hipGetErrorString(e));
    if (st != AH_OK) {
      ah_out_free(ctx, ov, std::max<size_t>(vbytes, 8));
      ah_out_free(ctx, ob, bbytes);
      ah_out_free(ctx, oo, obytes);
      return st;
    }
    out->values = ov;
    out->values_bytes = (int64_t)vbytes;
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->offsets = oo;
    out->offsets_bytes = (int64_t)obytes;
    out->null_count = len;
    return AH_OK;
  }
  const int64_t k = offset < 0 ? -offset : offset, keep = len - k, first = offset > 0 ? 0 : k;
  // the null piece
  const size_t zbytes = std::max<size_t>({is_str ? (size_t)(k + 1) * ow : (w ? (size_t)k * w : ah_bitmap_bytes(k)), ah_bitmap_bytes(k), (size_t)8});
  void* zeros = nullptr;
  AH_TRY(ah_pool_alloc(ctx, zbytes, &zeros));
  hipMemsetAsync(zeros, 0, zbytes, ctx->stream);
  ah_array_view nullp{};
  nullp.type = t;
  nullp.length = k;
  nullp.null_count = k;
  nullp.values = zeros;
  nullp.validity = (const uint8_t*)zeros;
  nullp.offsets = is_str ? zeros : nullptr;
  // the data piece: Array::slice(first, keep)
  ah_array_view data = *values;
  data.length = keep;
  data.null_count = -1;
  if (is_str) data.offsets = (const uint8_t*)values->offsets + (size_t)first * ow;
  else if (w == 0) data.values_bit_offset = values->values_bit_offset + first;
  else data.values = (const uint8_t*)values->values + (size_t)first * w;
  if (values->validity) data.validity_bit_offset = values->validity_bit_offset + first;
  else data.null_count = 0;
  ah_array_view pieces[2];
  pieces[0] = offset > 0 ? nullp : data;
  pieces[1] = offset > 0 ? data : nullp;
  const ah_status st = ah_concat(ctx, 2, pieces, out);
  ah_pool_free(ctx, zeros);  // the pool is stream-ordered: safe right after the enqueue
  return st;
}
