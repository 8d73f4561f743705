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

extern "C" ah_status ah_concat(ah_context* ctx, int32_t n, const ah_array_view* pieces,
                               ah_array_out* out) {
  if (!ctx || !out || (n > 0 && !pieces)) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (n <= 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "concat requires input of at least one array");
  const ah_type t = pieces[0].type;
  const int w = ah_type_width(t);
  if (w < 0 || t == AH_UTF8_VIEW || t == AH_BINARY_VIEW)  // views: buffer indices would need renumbering
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "concat not supported for type %s", ah_type_name(t));
  int64_t total = 0;
  bool any_nulls = false;
  for (int i = 0; i < n; ++i) {
    if (pieces[i].type != t)
      return ah_fail(ctx, AH_INVALID_ARGUMENT,
                     "It is not possible to concatenate arrays of different data types (%s, %s).",
                     ah_type_name(t), ah_type_name(pieces[i].type));
    total += pieces[i].length;
    int64_t nulls = 0;
    AH_TRY(ah_resolve_null_count(ctx, &pieces[i], &nulls));
    if (pieces[i].validity && nulls > 0) any_nulls = true;
  }
  out->type = t;
  out->length = total;
  if (total == 0) return AH_OK;
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
  hipError_t e = hipStreamSynchronize(ctx->stream);
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
  if (!src->validity) AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (appended_nulls) *appended_nulls = len - set;
  return AH_OK;
}
