// boolean.hip — arrow_arith::boolean on MI355X (predicate construction, so
// `lt(col, scalar) -> and -> filter` stays in HBM end to end).
//
// Reference: and / or / and_not (arrow-arith/src/boolean.rs:256-300, via binary_boolean_kernel :224:
// values = op(values), nulls = NullBuffer::union), not (:310, nulls cloned), and_kleene (:60-151),
// or_kleene (:156-222), is_null / is_not_null (:327-360, never carry a null buffer).
// Everything is word-parallel bitmap algebra (bitmap.hip), one thread per 64 result rows, inputs
// funnel-shifted from their own bit offsets.
#include "common.hpp"

extern "C" ah_status ah_boolean_binary(ah_context* ctx, ah_boolean_op op, const ah_array_view* l,
                                       const ah_array_view* r, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !l || !r || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_BOOL_AND || op > AH_BOOL_OR_KLEENE) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown boolean op %d", op);
  if (l->type != AH_BOOL || r->type != AH_BOOL)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "boolean kernels need Boolean inputs, got %s and %s",
                   ah_type_name(l->type), ah_type_name(r->type));
  if (l->length != r->length)
    return ah_fail(ctx, AH_COMPUTE_ERROR, "Cannot perform bitwise operation on arrays of different length");
  const int64_t len = l->length;
  out->type = AH_BOOL;
  out->length = len;
  if (len == 0) return AH_OK;
  const size_t bytes = ah_bitmap_bytes(len);
  const BitView none{nullptr, 0};
  BitView lv = make_bitview(l->values, l->values_bit_offset), rv = make_bitview(r->values, r->values_bit_offset);
  BitView ln = l->validity ? make_bitview(l->validity, l->validity_bit_offset) : none;
  BitView rn = r->validity ? make_bitview(r->validity, r->validity_bit_offset) : none;
  unsigned long long* vals = nullptr;
  unsigned long long* nb = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, (void**)&vals));
  int vop = (op == AH_BOOL_AND || op == AH_BOOL_AND_KLEENE) ? BM_AND : (op == AH_BOOL_AND_NOT ? BM_ANDNOT : BM_OR);
  ah_status st;
  {
    ah_prof_scope ps(ctx, "boolean");
    st = ah_bitmap_op(ctx, vop, lv, rv, none, len, vals, nullptr);
  }
  int64_t set = len;
  const bool has_nb = l->validity || r->validity;  // presence-based, like the reference
  if (st == AH_OK && has_nb) {
    st = ah_out_alloc(ctx, bytes, (void**)&nb);
    if (st == AH_OK) {
      if (op == AH_BOOL_AND_KLEENE || op == AH_BOOL_OR_KLEENE) {
        const bool is_and = op == AH_BOOL_AND_KLEENE;
        if (l->validity && r->validity)
          st = ah_bitmap_op(ctx, is_and ? BM_KLEENE_AND_NULLS : BM_KLEENE_OR_NULLS, ln, lv, rn, len, nb, AH_COUNT(ctx, &set), rv);
        else if (l->validity)  // nulls(left) | !values(right)   resp.   nulls(left) | values(right)
          st = ah_bitmap_op(ctx, is_and ? BM_OR_NOTB : BM_OR, ln, rv, none, len, nb, AH_COUNT(ctx, &set));
        else
          st = ah_bitmap_op(ctx, is_and ? BM_OR_NOTB : BM_OR, rn, lv, none, len, nb, AH_COUNT(ctx, &set));
      } else {
        st = ah_bitmap_op(ctx, (l->validity && r->validity) ? BM_AND : BM_COPY, l->validity ? ln : rn, rn, none,
                          len, nb, AH_COUNT(ctx, &set));
      }
    }
  }
  hipError_t e = hipSuccess;
  if (st == AH_OK) e = ah_end_of_call_sync(ctx);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, vals, bytes);
    ah_out_free(ctx, nb, bytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "boolean kernel failed: %s", hipGetErrorString(e));
  }
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (has_nb) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bytes;
    out->null_count = ah_nulls(ctx, len, set);
  }
  return AH_OK;
}

extern "C" ah_status ah_boolean_unary(ah_context* ctx, ah_boolean_op op, const ah_array_view* v,
                                      ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !v || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_BOOL_NOT || op > AH_BOOL_IS_NOT_NULL) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown boolean op %d", op);
  if (op == AH_BOOL_NOT && v->type != AH_BOOL)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "not() needs a Boolean input, got %s", ah_type_name(v->type));
  const int64_t len = v->length;
  out->type = AH_BOOL;
  out->length = len;
  if (len == 0) return AH_OK;
  const size_t bytes = ah_bitmap_bytes(len);
  const BitView none{nullptr, 0};
  unsigned long long* vals = nullptr;
  unsigned long long* nb = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, (void**)&vals));
  ah_status st = AH_OK;
  int64_t set = len;
  bool has_nb = false;
  if (op == AH_BOOL_NOT) {
    st = ah_bitmap_op(ctx, BM_NOT, make_bitview(v->values, v->values_bit_offset), none, none, len, vals, nullptr);
    if (st == AH_OK && v->validity) {  // nulls cloned
      has_nb = true;
      st = ah_out_alloc(ctx, bytes, (void**)&nb);
      if (st == AH_OK)
        st = ah_bitmap_op(ctx, BM_COPY, make_bitview(v->validity, v->validity_bit_offset), none, none, len, nb, AH_COUNT(ctx, &set));
    }
  } else if (!v->validity) {  // logical_nulls() == None
    if (op == AH_BOOL_IS_NULL) hipMemsetAsync(vals, 0, bytes, ctx->stream);
    else st = ah_bitmap_op(ctx, BM_COPY, none, none, none, len, vals, nullptr);  // new_set(len)
  } else {
    st = ah_bitmap_op(ctx, op == AH_BOOL_IS_NULL ? BM_NOT : BM_COPY,
                      make_bitview(v->validity, v->validity_bit_offset), none, none, len, vals, nullptr);
  }
  hipError_t e = hipSuccess;
  if (st == AH_OK) e = ah_end_of_call_sync(ctx);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, vals, bytes);
    ah_out_free(ctx, nb, bytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "boolean kernel failed: %s", hipGetErrorString(e));
  }
  out->values = vals;
  out->values_bytes = (int64_t)bytes;
  if (has_nb) {
    out->validity = (uint8_t*)nb;
    out->validity_bytes = (int64_t)bytes;
    out->null_count = ah_nulls(ctx, len, set);
  }
  return AH_OK;
}

// arrow_select::nullif::nullif (arrow-select/src/nullif.rs:60-121): the values (and offsets) are
// shared with `left` (zero copy, AH_OUT_BORROWED_VALUES); only the null buffer is new:
// validity = left_validity & !(right_values & right_validity), always present.
extern "C" ah_status ah_nullif(ah_context* ctx, const ah_array_view* left, const ah_array_view* right,
                               ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !left || !right || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (right->type != AH_BOOL)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "nullif needs a Boolean right-hand side, got %s", ah_type_name(right->type));
  if (left->length != right->length)
    return ah_fail(ctx, AH_COMPUTE_ERROR, "Cannot perform comparison operation on arrays of different length");
  const int64_t len = left->length;
  out->type = left->type;
  out->length = len;
  out->values = const_cast<void*>(left->values);
  out->values_bit_offset = left->values_bit_offset;
  out->offsets = const_cast<void*>(left->offsets);
  out->flags = AH_OUT_BORROWED_VALUES;
  if (len == 0) {  // make_array(left_data) unchanged
    out->flags = AH_OUT_BORROWED;
    out->validity = const_cast<uint8_t*>(left->validity);
    out->validity_bit_offset = left->validity_bit_offset;
    return AH_OK;
  }
  const size_t bytes = ah_bitmap_bytes(len);
  unsigned long long* nb = nullptr;
  AH_TRY(ah_out_alloc(ctx, bytes, (void**)&nb));
  const BitView none{nullptr, 0};
  int64_t set = 0;
  ah_status st = ah_bitmap_op(ctx, BM_NULLIF, left->validity ? make_bitview(left->validity, left->validity_bit_offset) : none,
                              make_bitview(right->values, right->values_bit_offset),
                              right->validity ? make_bitview(right->validity, right->validity_bit_offset) : none,
                              len, nb, AH_COUNT(ctx, &set));
  if (st != AH_OK) {
    ah_out_free(ctx, nb, bytes);
    ah_out_init(out);
    return st;
  }
  out->validity = (uint8_t*)nb;
  out->validity_bytes = (int64_t)bytes;
  out->null_count = ah_nulls(ctx, len, set);
  return AH_OK;
}
