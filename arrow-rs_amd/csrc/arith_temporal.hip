// arith_temporal.hip — arrow_arith::numeric::{add, sub, ..} when an operand is a temporal logical type:
// the type rules of `arithmetic_op` (arrow-arith/src/numeric.rs:225-275), `timestamp_op` (:426-537), `duration_op`
// (:877-895) and the Date - Date arms of `date_op` (:898-932).
//
// Every arm that is built here is checked i64 arithmetic under a different RESULT TYPE (Timestamp - Timestamp is a
// Duration; Timestamp +- Duration keeps the left zone; the *_wrapping entry points are checked too: the reference
// calls add_checked / sub_checked whatever the Op), so this file holds no kernel: it is the reference's `match` over
// (lhs type, op, rhs type), routed onto arith.hip's checked kernels — one pass over the operands, HBM-bound
// (24 B/row + validity) — plus the reference's InvalidArgumentError texts for everything `arithmetic_op` refuses.
// Date32 - Date32 (`((l as i64) - (r as i64)) * 86400`, infallible) is composed from cast -> sub_wrapping ->
// mul_wrapping by a scalar.  The Interval arms (calendar month arithmetic, :454-525, :934-961) are not built:
// AH_NOT_YET_IMPLEMENTED.
#include "common.hpp"
#include "temporal_cast.hpp"

ah_status ah_decimal_arith(ah_context* ctx, ah_arith_op op, const char* op_sym, const ah_array_view* lhs, int32_t l_s,
                           const ah_data_type* lt, const ah_array_view* rhs, int32_t r_s, const ah_data_type* rt,
                           ah_array_out* out, ah_data_type* out_type);  // arith_decimal.hip

namespace {

using namespace tc;

const char* op_text(ah_arith_op op) {  // Display for Op (numeric.rs:203-213)
  switch (op) {
    case AH_ADD: case AH_ADD_WRAPPING: return "+";
    case AH_SUB: case AH_SUB_WRAPPING: return "-";
    case AH_MUL: case AH_MUL_WRAPPING: return "*";
    case AH_DIV: return "/";
    default: return "%";
  }
}
bool is_add(ah_arith_op op) { return op == AH_ADD || op == AH_ADD_WRAPPING; }
bool is_sub(ah_arith_op op) { return op == AH_SUB || op == AH_SUB_WRAPPING; }
bool commutative(ah_arith_op op) { return is_add(op) || op == AH_MUL || op == AH_MUL_WRAPPING; }  // numeric.rs:215-222

std::string arith_type_text(const ah_data_type& t) {
  if (t.id == AH_DT_DECIMAL128) {
    char b[48];
    snprintf(b, sizeof b, "Decimal128(%d, %d)", t.precision, t.scale);
    return b;
  }
  if (t.id == AH_DT_INTERVAL) {
    static const char* n[] = {"YearMonth", "DayTime", "MonthDayNano"};
    return std::string("Interval(") + ((t.unit >= 0 && t.unit < 3) ? n[t.unit] : "?") + ")";
  }
  return type_text(t);
}

ah_data_type make_type(int32_t id, int32_t unit) {
  ah_data_type t{};
  t.id = id;
  t.unit = unit;
  return t;
}

struct Owned {  // a library-owned intermediate, released on every exit path
  ah_context* ctx;
  ah_array_out out;
  explicit Owned(ah_context* c) : ctx(c) { ah_out_init(&out); }
  ~Owned() { ah_array_release(ctx, &out); }
  ah_array_view view() const {
    ah_array_view v{};
    v.type = out.type;
    v.length = out.length;
    v.null_count = out.validity ? out.null_count : 0;
    v.values = out.values;
    v.validity = out.validity;
    v.validity_bit_offset = out.validity_bit_offset;
    return v;
  }
};

// op_ref!(DurationSecondType, l, l_s, r, r_s, ((l as i64) - (r as i64)) * NUM_SECONDS_IN_DAY)  (numeric.rs:913-924)
ah_status date32_diff(ah_context* ctx, const ah_array_view* l, int32_t l_s, const ah_array_view* r, int32_t r_s,
                      ah_array_out* out) {
  Owned l64(ctx), r64(ctx), diff(ctx);
  AH_TRY(ah_cast(ctx, l, AH_INT64, /*safe=*/0, &l64.out));  // try_unary: the nulls are cloned, nothing can fail
  AH_TRY(ah_cast(ctx, r, AH_INT64, 0, &r64.out));
  ah_array_view lv = l64.view(), rv = r64.view();
  AH_TRY(ah_arith_binary(ctx, AH_SUB_WRAPPING, &lv, l_s, &rv, r_s, &diff.out));
  void* k = nullptr;
  AH_TRY(ah_pool_alloc(ctx, 8, &k));
  const int64_t seconds_in_day = 86400;
  hipError_t e = hipMemcpyAsync(k, &seconds_in_day, 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = ah_stream_wait(ctx);  // the host constant must outlive the copy
  if (e != hipSuccess) {
    ah_pool_free(ctx, k);
    return ah_fail(ctx, AH_HIP_ERROR, "date32 difference: %s", hipGetErrorString(e));
  }
  ah_array_view kv{};
  kv.type = AH_INT64;
  kv.length = 1;
  kv.values = k;
  ah_array_view dv = diff.view();
  ah_status st = ah_arith_binary(ctx, AH_MUL_WRAPPING, &dv, 0, &kv, 1, out);
  ah_pool_free(ctx, k);
  return st;
}

}  // namespace

extern "C" ah_status ah_arith_with_types(ah_context* ctx, ah_arith_op op, const ah_array_view* lhs, int32_t l_s,
                                         const ah_data_type* lt, const ah_array_view* rhs, int32_t r_s,
                                         const ah_data_type* rt, ah_array_out* out, ah_data_type* out_type) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !lhs || !rhs || !lt || !rt || !out || !out_type) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (op < AH_ADD || op > AH_REM) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown arithmetic op %d", op);
  const int32_t L = lt->id, R = rt->id;
  auto logical = [](int32_t id) { return is_temporal(id) || id == AH_DT_INTERVAL || id == AH_DT_DECIMAL128; };
  if (!logical(L) && !logical(R)) {
    *out_type = *lt;
    return ah_arith_binary(ctx, op, lhs, l_s, rhs, r_s, out);
  }
  if ((is_temporal(L) && lhs->type != physical_of(*lt)) || (is_temporal(R) && rhs->type != physical_of(*rt)))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "operand layout does not match its logical type (%s as %s, %s as %s)",
                   arith_type_text(*lt).c_str(), ah_type_name(lhs->type), arith_type_text(*rt).c_str(), ah_type_name(rhs->type));
  // (Decimal128(_, _), Decimal128(_, _)) => decimal_op  (numeric.rs:259)
  if (L == AH_DT_DECIMAL128 && R == AH_DT_DECIMAL128) return ah_decimal_arith(ctx, op, op_text(op), lhs, l_s, lt, rhs, r_s, rt, out, out_type);
  const std::string ls = arith_type_text(*lt), rs = arith_type_text(*rt);
  auto checked_i64 = [&](ah_arith_op checked_op, const ah_data_type& result) {
    *out_type = result;
    return ah_arith_binary(ctx, checked_op, lhs, l_s, rhs, r_s, out);
  };
  auto interval_nyi = [&] {
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "%s %s %s: interval arithmetic is not built on the device", ls.c_str(),
                   op_text(op), rs.c_str());
  };
  if (L == AH_DT_TIMESTAMP) {  // timestamp_op, numeric.rs:426-537
    if (is_sub(op) && R == AH_DT_TIMESTAMP && rt->unit == lt->unit) return checked_i64(AH_SUB, make_type(AH_DT_DURATION, lt->unit));
    if (R == AH_DT_DURATION && rt->unit == lt->unit && (is_add(op) || is_sub(op))) return checked_i64(is_add(op) ? AH_ADD : AH_SUB, *lt);
    if (R == AH_DT_INTERVAL && (is_add(op) || is_sub(op))) return interval_nyi();
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid timestamp arithmetic operation: %s %s %s", ls.c_str(), op_text(op), rs.c_str());
  }
  if (L == AH_DT_DURATION && R == AH_DT_DURATION && lt->unit == rt->unit) {  // duration_op, :877-895
    if (is_add(op) || is_sub(op)) return checked_i64(is_add(op) ? AH_ADD : AH_SUB, *lt);
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid duration arithmetic operation: %s %s %s", ls.c_str(), op_text(op), rs.c_str());
  }
  if (L == AH_DT_INTERVAL && (R == AH_DT_INTERVAL ? rt->unit == lt->unit : (R == AH_INT64 || (R == AH_FLOAT64 && lt->unit == 2))))
    return interval_nyi();  // interval_op / interval_f64_op, :819-875
  if (L == AH_DT_DATE32 || L == AH_DT_DATE64) {  // date_op, :898-969
    if (is_sub(op) && R == L) {
      if (L == AH_DT_DATE64) return checked_i64(AH_SUB, make_type(AH_DT_DURATION, AH_MILLISECOND));
      *out_type = make_type(AH_DT_DURATION, AH_SECOND);
      return date32_diff(ctx, lhs, l_s, rhs, r_s, out);
    }
    if (R == AH_DT_INTERVAL && (is_add(op) || is_sub(op))) return interval_nyi();
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid date arithmetic operation: %s %s %s", ls.c_str(), op_text(op), rs.c_str());
  }
  // the last arm (:262-273): Duration / Interval on the left of a commutative op swaps; Int64 * Interval swaps
  if ((L == AH_DT_DURATION || L == AH_DT_INTERVAL) && (R == AH_DT_DATE32 || R == AH_DT_DATE64 || R == AH_DT_TIMESTAMP) && commutative(op))
    return ah_arith_with_types(ctx, op, rhs, r_s, rt, lhs, l_s, lt, out, out_type);
  if (((L == AH_INT64 && R == AH_DT_INTERVAL) || (L == AH_FLOAT64 && R == AH_DT_INTERVAL && rt->unit == 2)) && op == AH_MUL)
    return interval_nyi();
  return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid arithmetic operation: %s %s %s", ls.c_str(), op_text(op), rs.c_str());
}

// compare_op's type rule behind the C ABI (arrow-ord/src/cmp.rs:243-264): the LOGICAL types of the two sides must be equal —
// Decimal128(12, 3) vs Decimal128(12, 1), Timestamp(Second) vs Timestamp(Millisecond), Date32 vs Int32 are refused with the
// reference's text — then the values compare as their physical type (ah_compare).  Round 4 kept this check in the Python
// mirror (compute/kernels/cmp.py), where a Rust host would have had to re-implement it.
extern "C" ah_status ah_compare_with_types(ah_context* ctx, ah_cmp_op op, const ah_array_view* lhs, int32_t lhs_is_scalar,
                                           const ah_data_type* lhs_type, const ah_array_view* rhs, int32_t rhs_is_scalar,
                                           const ah_data_type* rhs_type, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !lhs || !rhs || !lhs_type || !rhs_type || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  if (op < AH_EQ || op > AH_NOT_DISTINCT) return ah_fail(ctx, AH_INVALID_ARGUMENT, "unknown comparison op %d", op);
  const ah_data_type &l = *lhs_type, &r = *rhs_type;
  const bool same = l.id == r.id && l.unit == r.unit && (l.has_tz != 0) == (r.has_tz != 0) &&
                    (!l.has_tz || l.tz_offset_seconds == r.tz_offset_seconds) && l.precision == r.precision && l.scale == r.scale;
  if (!same) {
    static const char* sym[] = {"==", "!=", "<", "<=", ">", ">=", "IS DISTINCT FROM", "IS NOT DISTINCT FROM"};  // Display for Op (cmp.rs:55-68)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Invalid comparison operation: %s %s %s", arith_type_text(l).c_str(), sym[op - AH_EQ],
                   arith_type_text(r).c_str());
  }
  return ah_compare(ctx, op, lhs, lhs_is_scalar, rhs, rhs_is_scalar, out);
}
