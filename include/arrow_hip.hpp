// arrow_hip.hpp — C++17 host-side mirror of `arrow::compute::kernels` over the C ABI.
//
// The reference is compiled (Rust) code and no Rust toolchain exists in this image, so the
// host side above the C ABI is C++: same function names, argument order, option structs and
// error behaviour as the reference (`Result<ArrayRef, ArrowError>` becomes a thrown
// `ArrowError`, a reference panic becomes a thrown `Panic`).  Header-only; needs only
// arrow_hip.h and libarrow_hip.so (no HIP headers, no torch).
//
//   arrow_hip::compute::filter            arrow-select/src/filter.rs:201
//   arrow_hip::compute::filter_record_batch                        :225
//   arrow_hip::compute::FilterBuilder / FilterPredicate            :248-533
//   arrow_hip::compute::Term / filter_expr / FilterBuilder::from_terms
//                                         the lazy form of filter(v, and_kleene(lt(..), gt_eq(..))): cmp.rs:113-164,
//                                         arrow-arith/src/boolean.rs:60-300, filter.rs:201 (ah_filter_expr)
//   arrow_hip::compute::take / TakeOptions arrow-select/src/take.rs:89, :388
//   arrow_hip::compute::{add,add_wrapping,sub,...,rem,neg,neg_wrapping}
//                                         arrow-arith/src/numeric.rs:36-186
//   arrow_hip::compute::{eq,neq,lt,lt_eq,gt,gt_eq,distinct,not_distinct}
//                                         arrow-ord/src/cmp.rs:79-202
//   arrow_hip::compute::{and_,or_,and_not,and_kleene,or_kleene,not_,is_null,is_not_null}
//                                         arrow-arith/src/boolean.rs:60-360
//   arrow_hip::compute::cast / cast_with_options / CastOptions
//                                         arrow-cast/src/cast/mod.rs:347,:790,:95
//   arrow_hip::compute::concat            arrow-select/src/concat.rs:495
#pragma once
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "arrow_hip.h"

namespace arrow_hip {

// ArrowError (arrow-schema/src/error.rs:26-69); what() == Rust Display
class ArrowError : public std::runtime_error {
 public:
  ArrowError(ah_status code, std::string msg)
      : std::runtime_error(prefix(code) + (code == AH_DIVIDE_BY_ZERO ? std::string() : msg)),
        code_(code), message_(std::move(msg)) {}
  ah_status code() const { return code_; }
  const std::string& message() const { return message_; }

 private:
  static std::string prefix(ah_status c) {
    switch (c) {
      case AH_INVALID_ARGUMENT: return "Invalid argument error: ";
      case AH_COMPUTE_ERROR: return "Compute error: ";
      case AH_ARITHMETIC_OVERFLOW: return "Arithmetic overflow: ";
      case AH_DIVIDE_BY_ZERO: return "Divide by zero error";
      case AH_CAST_ERROR: return "Cast error: ";
      case AH_NOT_YET_IMPLEMENTED: return "Not yet implemented: ";
      case AH_OFFSET_OVERFLOW_ERROR: return "Offset overflow error: ";
      case AH_C_DATA_INTERFACE: return "C Data interface error: ";
      case AH_IPC_ERROR: return "Ipc error: ";
      case AH_PARSE_ERROR: return "Parser error: ";
      default: return "";
    }
  }
  ah_status code_;
  std::string message_;
};
// the reference would panic!() (OOB take without check_bounds, offset overflow)
class Panic : public std::runtime_error {
  using std::runtime_error::runtime_error;
};

// One HIP stream + pooled HBM allocator.  Entry points lock their context for the duration of a call, so a Context may
// be shared by threads (calls serialise); threads that want overlap use one each.
class Context {
 public:
  explicit Context(int device = 0) {
    if (ah_context_create(device, &h_) != AH_OK)
      throw std::runtime_error("ah_context_create failed: no usable MI355X (no CPU fallback)");
  }
  ~Context() { ah_context_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  ah_context* handle() const { return h_; }
  // opt-in asynchronous mode (ah_context_set_deferred): infallible fixed-shape kernels only enqueue;
  // Array::null_count() of such a result counts lazily
  void set_deferred(bool on) { ah_context_set_deferred(h_, on ? 1 : 0); }
  bool deferred() const { return ah_context_deferred(h_) != 0; }
  void synchronize() const { check(ah_synchronize(h_)); }
  // hipGraph capture of deferred calls (ah_graph_begin / _end): `auto g = ctx.capture([&] { ... deferred kernels ... });`
  // records the calls the callable makes instead of running them; g->launch() replays them as one graph launch over the
  // current bytes of the captured inputs, into the outputs the recorded calls returned (keep those alive)
  class Graph {
   public:
    Graph(const Context* c, ah_graph* g) : c_(c), g_(g) {}
    Graph(const Graph&) = delete;
    ~Graph() { ah_graph_destroy(c_->h_, g_); }
    void launch() const { c_->check(ah_graph_launch(c_->h_, g_)); }
    int node_count() const { return ah_graph_node_count(g_); }

   private:
    const Context* c_;
    ah_graph* g_;
  };
  template <typename F>
  std::unique_ptr<Graph> capture(F&& record) const {
    check(ah_graph_begin(h_));
    ah_graph* g = nullptr;
    try {
      record();
    } catch (...) {
      if (ah_graph_end(h_, &g) == AH_OK) ah_graph_destroy(h_, g);
      throw;
    }
    check(ah_graph_end(h_, &g));
    return std::make_unique<Graph>(this, g);
  }
  // MemoryPool::used and friends for the pooled device allocator (ah_context_stats)
  ah_context_stats_t memory_stats(bool reset_peaks = false) const {
    ah_context_stats_t st{};
    check(ah_context_stats(h_, &st, reset_peaks ? 1 : 0));
    return st;
  }
  void check(ah_status st) const {
    if (st == AH_OK) return;
    std::string m = ah_last_error(h_);
    if (st == AH_PANIC || st == AH_OFFSET_OVERFLOW) throw Panic(m);
    throw ArrowError(st, m);
  }

 private:
  ah_context* h_ = nullptr;
};

inline int device_count() { return (int)ah_device_count(); }  // GPUs visible to this process (hipGetDeviceCount)

// An array whose buffers live in HBM (PrimitiveArray<T> / BooleanArray / StringArray).
// Owns an ah_array_out, or borrows a caller-described view.
class Array {
 public:
  Array(std::shared_ptr<Context> ctx, const ah_array_out& out, std::vector<std::shared_ptr<Array>> keep = {})
      : ctx_(std::move(ctx)), out_(out), owned_(true), keep_(std::move(keep)) {
    view_.type = out.type;
    view_.length = out.length;
    view_.null_count = out.validity ? out.null_count : 0;
    view_.values = out.values;
    view_.values_bit_offset = out.values_bit_offset;
    view_.validity = out.validity;
    view_.validity_bit_offset = out.validity_bit_offset;
    view_.offsets = out.offsets;
  }
  // wrap device memory the caller owns (kept alive by the caller)
  Array(std::shared_ptr<Context> ctx, const ah_array_view& v) : ctx_(std::move(ctx)), view_(v) {
    std::memset(&out_, 0, sizeof out_);
  }
  ~Array() {
    if (owned_) ah_array_release(ctx_->handle(), &out_);
  }
  Array(const Array&) = delete;
  Array& operator=(const Array&) = delete;

  ah_type data_type() const { return view_.type; }
  int64_t len() const { return view_.length; }
  bool is_empty() const { return view_.length == 0; }
  int64_t null_count() const {
    if (!view_.validity) return 0;
    if (view_.null_count < 0) {  // deferred result: counted (and the stream synchronised) on first use
      int64_t set = 0;
      ctx_->check(ah_count_set_bits(ctx_->handle(), view_.validity, view_.validity_bit_offset, view_.length, &set));
      view_.null_count = view_.length - set;
    }
    return view_.null_count;
  }
  bool has_nulls_buffer() const { return view_.validity != nullptr; }  // nulls().is_some()
  const ah_array_view& view() const { return view_; }
  const void* offsets() const { return out_.offsets; }
  const std::shared_ptr<Context>& context() const { return ctx_; }
  // Buffer::shrink_to_fit (arrow-buffer/src/buffer/immutable.rs:215): results of small-batch filters are allocated for the
  // worst case (K is not known at allocation time); copies them into exact-size buffers.  Owned fixed-width results only.
  void shrink_to_fit() {
    if (!owned_) return;
    ctx_->check(ah_array_shrink_to_fit(ctx_->handle(), &out_));
    view_.values = out_.values;
    view_.validity = out_.validity;
  }
  int64_t values_capacity_bytes() const { return out_.values_bytes; }
  // copy `bytes` of the values buffer to the host (debug / tests)
  void values_to_host(void* dst, size_t bytes) const {
    ctx_->check(ah_memcpy_dtoh(ctx_->handle(), dst, view_.values, bytes));
  }

 private:
  std::shared_ptr<Context> ctx_;
  mutable ah_array_view view_{};
  ah_array_out out_{};
  bool owned_ = false;
  std::vector<std::shared_ptr<Array>> keep_;
};
using ArrayRef = std::shared_ptr<Array>;

// Datum (arrow-array/src/scalar.rs:78-98): an array, or a length-1 array used as a scalar
struct Datum {
  ArrayRef array;
  bool is_scalar = false;
  Datum(ArrayRef a) : array(std::move(a)) {}  // NOLINT: arrays convert implicitly, like &dyn Datum
  Datum(ArrayRef a, bool scalar) : array(std::move(a)), is_scalar(scalar) {}
};
inline Datum Scalar(ArrayRef one_element_array) { return Datum(std::move(one_element_array), true); }

struct RecordBatch {  // arrow-array/src/record_batch.rs:224 (columns only)
  std::vector<ArrayRef> columns;
  int64_t num_rows = 0;
};

namespace compute {

inline ArrayRef wrap(const ArrayRef& like, const ah_array_out& out, std::vector<ArrayRef> keep = {}) {
  return std::make_shared<Array>(like->context(), out, std::move(keep));
}

// ---- filter (arrow-select/src/filter.rs)
inline ArrayRef filter(const ArrayRef& values, const ArrayRef& predicate) {
  ah_array_out out;
  values->context()->check(ah_filter(values->context()->handle(), &values->view(), &predicate->view(), &out));
  return wrap(values, out, {values});
}

class FilterPredicate {  // filter.rs:442-533
 public:
  FilterPredicate(ArrayRef predicate, ah_filter_predicate* p) : predicate_(std::move(predicate)), p_(p) {}
  ~FilterPredicate() { ah_filter_predicate_free(predicate_->context()->handle(), p_); }
  FilterPredicate(const FilterPredicate&) = delete;
  int64_t count() const { return ah_filter_predicate_count(p_); }
  ArrayRef filter(const ArrayRef& values) const {
    ah_array_out out;
    values->context()->check(
        ah_filter_predicate_apply(values->context()->handle(), p_, &values->view(), &out));
    return wrap(values, out, {values});
  }
  RecordBatch filter_record_batch(const RecordBatch& rb) const {
    RecordBatch o;
    for (auto& c : rb.columns) o.columns.push_back(filter(c));
    o.num_rows = count();
    return o;
  }

 private:
  ArrayRef predicate_;
  ah_filter_predicate* p_;
};

// One comparison of a filter expression: `op(lhs, rhs)` with Datum operands (cmp.rs:79-202).
struct Term {
  ah_cmp_op op;
  Datum lhs, rhs;
};
namespace detail {
// the C form of an expression; the views point into the Arrays the terms hold
struct TermArrays {
  std::vector<ah_filter_term> terms;
  std::vector<int32_t> joins;
  std::vector<ArrayRef> keep;
  std::shared_ptr<Context> ctx;
  TermArrays(const std::vector<Term>& ts, const std::vector<ah_boolean_op>& js) {
    if (ts.empty()) throw ArrowError(AH_INVALID_ARGUMENT, "a filter expression needs at least one term");
    if (js.size() + 1 != ts.size()) throw ArrowError(AH_INVALID_ARGUMENT, "a filter expression of n terms takes n - 1 joins");
    ctx = ts[0].lhs.array->context();
    for (auto& t : ts) {
      ah_filter_term c{};
      c.op = (int32_t)t.op;
      c.lhs = &t.lhs.array->view(), c.lhs_is_scalar = t.lhs.is_scalar ? 1 : 0;
      c.rhs = &t.rhs.array->view(), c.rhs_is_scalar = t.rhs.is_scalar ? 1 : 0;
      terms.push_back(c);
      keep.push_back(t.lhs.array);
      keep.push_back(t.rhs.array);
    }
    for (auto j : js) joins.push_back((int32_t)j);
  }
};
}  // namespace detail

class FilterBuilder {  // filter.rs:248-324
 public:
  explicit FilterBuilder(ArrayRef predicate) : predicate_(std::move(predicate)) {}
  // The lazy predicate: what `FilterBuilder::new(&and_kleene(&lt(a, x)?, &gt_eq(b, y)?)?)` selects, with the comparisons
  // evaluated inside the filter's count pass instead of being materialised (ah_filter_predicate_build_expr).  Terms are
  // folded left to right by `joins` (AH_BOOL_AND / _OR / _AND_KLEENE / _OR_KLEENE).
  static FilterBuilder from_terms(std::vector<Term> terms, std::vector<ah_boolean_op> joins) {
    FilterBuilder b{ArrayRef()};
    b.terms_ = std::move(terms);
    b.joins_ = std::move(joins);
    return b;
  }
  FilterBuilder& optimize() { return *this; }  // device predicates always carry prefix tables
  std::unique_ptr<FilterPredicate> build() {
    ah_filter_predicate* p = nullptr;
    if (!predicate_) {
      detail::TermArrays t(terms_, joins_);
      t.ctx->check(ah_filter_predicate_build_expr(t.ctx->handle(), (int32_t)t.terms.size(), t.terms.data(), t.joins.data(), &p));
      return std::make_unique<FilterPredicate>(terms_[0].lhs.array, p);  // (the predicate owns its selection words)
    }
    predicate_->context()->check(
        ah_filter_predicate_build(predicate_->context()->handle(), &predicate_->view(), &p));
    return std::make_unique<FilterPredicate>(predicate_, p);
  }

 private:
  ArrayRef predicate_;
  std::vector<Term> terms_;
  std::vector<ah_boolean_op> joins_;
};

// `filter(values, <terms folded by joins>)` in one call (ah_filter_expr)
inline ArrayRef filter_expr(const ArrayRef& values, const std::vector<Term>& terms, const std::vector<ah_boolean_op>& joins) {
  detail::TermArrays t(terms, joins);
  ah_array_out out;
  t.ctx->check(ah_filter_expr(t.ctx->handle(), (int32_t)t.terms.size(), t.terms.data(), t.joins.data(), &values->view(), &out));
  return wrap(values, out, {values});
}

// filter_record_batch (filter.rs:225): ONE count pass and ONE scatter launch per group of same-shaped columns
// (ah_filter_record_batch); batches of at most 2^20 rows take the one-launch path
inline RecordBatch filter_record_batch(const RecordBatch& rb, const ArrayRef& predicate) {
  auto& ctx = predicate->context();
  std::vector<ah_array_view> views;
  for (auto& c : rb.columns) views.push_back(c->view());
  std::vector<ah_array_out> outs(views.size());
  int64_t rows = 0;
  ctx->check(ah_filter_record_batch(ctx->handle(), (int32_t)views.size(), views.data(), &predicate->view(), outs.data(), &rows));
  RecordBatch o;
  o.num_rows = rows;
  for (size_t i = 0; i < outs.size(); ++i) o.columns.push_back(wrap(rb.columns[i], outs[i], {rb.columns[i]}));
  return o;
}

// ---- take (arrow-select/src/take.rs)
struct TakeOptions {  // take.rs:388-394
  bool check_bounds = false;
};
inline ArrayRef take(const ArrayRef& values, const ArrayRef& indices, TakeOptions options = {}) {
  ah_array_out out;
  values->context()->check(ah_take(values->context()->handle(), &values->view(), &indices->view(),
                                   options.check_bounds ? 1 : 0, &out));
  return wrap(values, out);
}

// ---- numeric (arrow-arith/src/numeric.rs)
inline ArrayRef arith(ah_arith_op op, const Datum& l, const Datum& r) {
  ah_array_out out;
  auto& ctx = l.array->context();
  ctx->check(ah_arith_binary(ctx->handle(), op, &l.array->view(), l.is_scalar, &r.array->view(),
                             r.is_scalar, &out));
  return wrap(l.array, out);
}
inline ArrayRef add(const Datum& l, const Datum& r) { return arith(AH_ADD, l, r); }
inline ArrayRef add_wrapping(const Datum& l, const Datum& r) { return arith(AH_ADD_WRAPPING, l, r); }
inline ArrayRef sub(const Datum& l, const Datum& r) { return arith(AH_SUB, l, r); }
inline ArrayRef sub_wrapping(const Datum& l, const Datum& r) { return arith(AH_SUB_WRAPPING, l, r); }
inline ArrayRef mul(const Datum& l, const Datum& r) { return arith(AH_MUL, l, r); }
inline ArrayRef mul_wrapping(const Datum& l, const Datum& r) { return arith(AH_MUL_WRAPPING, l, r); }
inline ArrayRef div(const Datum& l, const Datum& r) { return arith(AH_DIV, l, r); }
inline ArrayRef rem(const Datum& l, const Datum& r) { return arith(AH_REM, l, r); }
inline ArrayRef neg(const ArrayRef& a) {
  ah_array_out out;
  a->context()->check(ah_arith_neg(a->context()->handle(), &a->view(), 0, &out));
  return wrap(a, out);
}
inline ArrayRef neg_wrapping(const ArrayRef& a) {
  ah_array_out out;
  a->context()->check(ah_arith_neg(a->context()->handle(), &a->view(), 1, &out));
  return wrap(a, out);
}

// ---- cmp (arrow-ord/src/cmp.rs)
inline ArrayRef compare(ah_cmp_op op, const Datum& l, const Datum& r) {
  ah_array_out out;
  auto& ctx = l.array->context();
  ctx->check(ah_compare(ctx->handle(), op, &l.array->view(), l.is_scalar, &r.array->view(), r.is_scalar, &out));
  return wrap(l.array, out);
}
inline ArrayRef eq(const Datum& l, const Datum& r) { return compare(AH_EQ, l, r); }
inline ArrayRef neq(const Datum& l, const Datum& r) { return compare(AH_NEQ, l, r); }
inline ArrayRef lt(const Datum& l, const Datum& r) { return compare(AH_LT, l, r); }
inline ArrayRef lt_eq(const Datum& l, const Datum& r) { return compare(AH_LT_EQ, l, r); }
inline ArrayRef gt(const Datum& l, const Datum& r) { return compare(AH_GT, l, r); }
inline ArrayRef gt_eq(const Datum& l, const Datum& r) { return compare(AH_GT_EQ, l, r); }
inline ArrayRef distinct(const Datum& l, const Datum& r) { return compare(AH_DISTINCT, l, r); }
inline ArrayRef not_distinct(const Datum& l, const Datum& r) { return compare(AH_NOT_DISTINCT, l, r); }

// ---- boolean (arrow-arith/src/boolean.rs); `and`/`or`/`not` are C++ alternative tokens -> trailing underscore
inline ArrayRef boolean_binary(ah_boolean_op op, const ArrayRef& l, const ArrayRef& r) {
  ah_array_out out;
  l->context()->check(ah_boolean_binary(l->context()->handle(), op, &l->view(), &r->view(), &out));
  return wrap(l, out);
}
inline ArrayRef boolean_unary(ah_boolean_op op, const ArrayRef& v) {
  ah_array_out out;
  v->context()->check(ah_boolean_unary(v->context()->handle(), op, &v->view(), &out));
  return wrap(v, out);
}
inline ArrayRef and_(const ArrayRef& l, const ArrayRef& r) { return boolean_binary(AH_BOOL_AND, l, r); }
inline ArrayRef or_(const ArrayRef& l, const ArrayRef& r) { return boolean_binary(AH_BOOL_OR, l, r); }
inline ArrayRef and_not(const ArrayRef& l, const ArrayRef& r) { return boolean_binary(AH_BOOL_AND_NOT, l, r); }
inline ArrayRef and_kleene(const ArrayRef& l, const ArrayRef& r) { return boolean_binary(AH_BOOL_AND_KLEENE, l, r); }
inline ArrayRef or_kleene(const ArrayRef& l, const ArrayRef& r) { return boolean_binary(AH_BOOL_OR_KLEENE, l, r); }
inline ArrayRef not_(const ArrayRef& v) { return boolean_unary(AH_BOOL_NOT, v); }
inline ArrayRef is_null(const ArrayRef& v) { return boolean_unary(AH_BOOL_IS_NULL, v); }
inline ArrayRef is_not_null(const ArrayRef& v) { return boolean_unary(AH_BOOL_IS_NOT_NULL, v); }
inline ArrayRef nullif(const ArrayRef& l, const ArrayRef& r) {  // arrow-select/src/nullif.rs:60
  ah_array_out out;
  l->context()->check(ah_nullif(l->context()->handle(), &l->view(), &r->view(), &out));
  return wrap(l, out, {l});
}

// ---- cast (arrow-cast/src/cast/mod.rs)
struct CastOptions {  // cast/mod.rs:95-111
  bool safe = true;
};
inline bool can_cast_types(ah_type from, ah_type to) { return ah_can_cast_types(from, to) != 0; }
inline ArrayRef cast_with_options(const ArrayRef& a, ah_type to, const CastOptions& o) {
  ah_array_out out;
  a->context()->check(ah_cast(a->context()->handle(), &a->view(), to, o.safe ? 1 : 0, &out));
  return wrap(a, out);
}
inline ArrayRef cast(const ArrayRef& a, ah_type to) { return cast_with_options(a, to, CastOptions{}); }
// The temporal arms (cast/mod.rs:1700-2260) need the logical types: DataType::Timestamp(unit, tz) etc. as ah_data_type.
inline ah_data_type plain_type(ah_type t) { return ah_data_type{t, 0, 0, 0, 0, 0}; }
inline ah_data_type date32() { return ah_data_type{AH_DT_DATE32, 0, 0, 0, 0, 0}; }
inline ah_data_type date64() { return ah_data_type{AH_DT_DATE64, 0, 0, 0, 0, 0}; }
inline ah_data_type time32(ah_time_unit u) { return ah_data_type{AH_DT_TIME32, u, 0, 0, 0, 0}; }
inline ah_data_type time64(ah_time_unit u) { return ah_data_type{AH_DT_TIME64, u, 0, 0, 0, 0}; }
inline ah_data_type duration(ah_time_unit u) { return ah_data_type{AH_DT_DURATION, u, 0, 0, 0, 0}; }
inline ah_data_type timestamp(ah_time_unit u) { return ah_data_type{AH_DT_TIMESTAMP, u, 0, 0, 0, 0}; }
inline ah_data_type timestamp(ah_time_unit u, int32_t tz_offset_seconds) {  // a fixed-offset zone, "+05:45" = 20700
  return ah_data_type{AH_DT_TIMESTAMP, u, 1, tz_offset_seconds, 0, 0};
}
inline bool can_cast_types(const ah_data_type& from, const ah_data_type& to) { return ah_can_cast_data_types(&from, &to) != 0; }
inline ArrayRef cast_with_options(const ArrayRef& a, const ah_data_type& from, const ah_data_type& to, const CastOptions& o) {
  ah_array_out out;
  a->context()->check(ah_cast_with_types(a->context()->handle(), &a->view(), &from, &to, o.safe ? 1 : 0, &out));
  return wrap(a, out);
}
inline ArrayRef cast(const ArrayRef& a, const ah_data_type& from, const ah_data_type& to) {
  return cast_with_options(a, from, to, CastOptions{});
}
// numeric.rs with temporal operands (timestamp_op :426, duration_op :877, date_op :898): the result's logical type
// is returned through `out_type` (Timestamp - Timestamp = Duration, Timestamp +- Duration keeps the zone, ...).
inline ArrayRef arith(ah_arith_op op, const Datum& l, const ah_data_type& lt, const Datum& r, const ah_data_type& rt,
                      ah_data_type* out_type) {
  ah_array_out out;
  auto& ctx = l.array->context();
  ctx->check(ah_arith_with_types(ctx->handle(), op, &l.array->view(), l.is_scalar, &lt, &r.array->view(), r.is_scalar, &rt,
                                 &out, out_type));
  return wrap(l.array, out);
}

// ---- concat (arrow-select/src/concat.rs)
inline ArrayRef concat(const std::vector<ArrayRef>& arrays) {
  if (arrays.empty()) throw ArrowError(AH_INVALID_ARGUMENT, "concat requires input of at least one array");
  std::vector<ah_array_view> views;
  for (auto& a : arrays) views.push_back(a->view());
  ah_array_out out;
  arrays[0]->context()->check(ah_concat(arrays[0]->context()->handle(), (int32_t)views.size(), views.data(), &out));
  return wrap(arrays[0], out);
}

// concat_batches (arrow-select/src/concat.rs:607): column i = concat of the batches' i-th columns; with no
// columns the row counts are summed (:612-617)
inline RecordBatch concat_batches(size_t n_fields, const std::vector<const RecordBatch*>& batches) {
  RecordBatch out;
  if (n_fields == 0) {
    for (auto* b : batches) out.num_rows += b->num_rows;
    return out;
  }
  if (batches.empty()) return out;
  for (size_t i = 0; i < n_fields; ++i) {
    std::vector<ArrayRef> col;
    for (auto* b : batches) col.push_back(b->columns.at(i));
    out.columns.push_back(concat(col));
  }
  out.num_rows = out.columns[0]->len();
  return out;
}

// ---- aggregate (arrow-arith/src/aggregate.rs): std::optional<T> mirrors Option<T::Native>
template <typename T> inline std::optional<T> aggregate(ah_agg_op op, const ArrayRef& a) {
  ah_scalar s;
  a->context()->check(ah_aggregate(a->context()->handle(), op, &a->view(), &s));
  if (!s.is_valid) return std::nullopt;
  T v;
  std::memcpy(&v, s.bytes, sizeof(T));
  return v;
}
template <typename T> inline std::optional<T> sum(const ArrayRef& a) { return aggregate<T>(AH_AGG_SUM, a); }
template <typename T> inline std::optional<T> sum_checked(const ArrayRef& a) { return aggregate<T>(AH_AGG_SUM_CHECKED, a); }
template <typename T> inline std::optional<T> product(const ArrayRef& a) { return aggregate<T>(AH_AGG_PRODUCT, a); }
template <typename T> inline std::optional<T> product_checked(const ArrayRef& a) { return aggregate<T>(AH_AGG_PRODUCT_CHECKED, a); }
template <typename T> inline std::optional<T> min(const ArrayRef& a) { return aggregate<T>(AH_AGG_MIN, a); }
template <typename T> inline std::optional<T> max(const ArrayRef& a) { return aggregate<T>(AH_AGG_MAX, a); }
template <typename T> inline std::optional<T> bit_and(const ArrayRef& a) { return aggregate<T>(AH_AGG_BIT_AND, a); }
template <typename T> inline std::optional<T> bit_or(const ArrayRef& a) { return aggregate<T>(AH_AGG_BIT_OR, a); }
template <typename T> inline std::optional<T> bit_xor(const ArrayRef& a) { return aggregate<T>(AH_AGG_BIT_XOR, a); }
inline std::optional<bool> min_boolean(const ArrayRef& a) { return aggregate<bool>(AH_AGG_MIN, a); }
inline std::optional<bool> max_boolean(const ArrayRef& a) { return aggregate<bool>(AH_AGG_MAX, a); }
inline std::optional<bool> bool_and(const ArrayRef& a) { return min_boolean(a); }
inline std::optional<bool> bool_or(const ArrayRef& a) { return max_boolean(a); }

// ---- bitwise (arrow-arith/src/bitwise.rs)
inline ArrayRef bitwise_and(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_AND, l, r); }
inline ArrayRef bitwise_or(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_OR, l, r); }
inline ArrayRef bitwise_xor(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_XOR, l, r); }
inline ArrayRef bitwise_shift_left(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_SHIFT_LEFT, l, r); }
inline ArrayRef bitwise_shift_right(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_SHIFT_RIGHT, l, r); }
inline ArrayRef bitwise_and_not(const Datum& l, const Datum& r) { return arith((ah_arith_op)AH_BIT_AND_NOT, l, r); }
inline ArrayRef bitwise_not(const ArrayRef& a) {
  ah_array_out out;
  a->context()->check(ah_bitwise_not(a->context()->handle(), &a->view(), &out));
  return wrap(a, out);
}

// ---- interleave (arrow-select/src/interleave.rs:74); indices = (array, row) pairs as two UInt32 device arrays
inline ArrayRef interleave(const std::vector<ArrayRef>& values, const ArrayRef& array_index, const ArrayRef& row_index) {
  std::vector<ah_array_view> views;
  for (auto& v : values) views.push_back(v->view());
  ah_array_out out;
  array_index->context()->check(ah_interleave(array_index->context()->handle(), (int32_t)views.size(), views.data(),
                                              &array_index->view(), &row_index->view(), &out));
  return std::make_shared<Array>(array_index->context(), out);
}

// ---- zip (arrow-select/src/zip.rs:99)
inline ArrayRef zip(const ArrayRef& mask, const Datum& truthy, const Datum& falsy) {
  ah_array_out out;
  mask->context()->check(ah_zip(mask->context()->handle(), &mask->view(), &truthy.array->view(), truthy.is_scalar,
                                &falsy.array->view(), falsy.is_scalar, &out));
  return wrap(truthy.array, out);
}

// ---- sort (arrow-ord/src/sort.rs)
struct SortOptions {  // arrow-schema/src/lib.rs:87; default ASC NULLS FIRST
  bool descending = false;
  bool nulls_first = true;
};
inline ArrayRef sort_to_indices(const ArrayRef& values, SortOptions options = {}, int64_t limit = -1) {
  ah_array_out out;
  values->context()->check(ah_sort_to_indices(values->context()->handle(), &values->view(), options.descending,
                                              options.nulls_first, limit, &out));
  return wrap(values, out);
}
struct SortColumn {  // sort.rs:870
  ArrayRef values;
  SortOptions options;
};
inline ArrayRef lexsort_to_indices(const std::vector<SortColumn>& columns, int64_t limit = -1) {
  if (columns.empty()) throw ArrowError(AH_INVALID_ARGUMENT, "Sort requires at least one column");
  std::vector<ah_array_view> views;
  std::vector<int32_t> desc, nf;
  for (auto& c : columns) {
    views.push_back(c.values->view());
    desc.push_back(c.options.descending);
    nf.push_back(c.options.nulls_first);
  }
  ah_array_out out;
  auto ctx = columns[0].values->context();
  ctx->check(ah_lexsort_to_indices(ctx->handle(), (int32_t)views.size(), views.data(), desc.data(), nf.data(), limit, &out));
  return std::make_shared<Array>(ctx, out);
}
// arrow_ord::rank::rank (rank.rs:58): UInt32 ranks, never null (the reference's Vec<u32>)
inline ArrayRef rank(const ArrayRef& values, SortOptions options = {}) {
  ah_array_out out;
  values->context()->check(ah_rank(values->context()->handle(), &values->view(), options.descending, options.nulls_first, &out));
  return wrap(values, out);
}
// arrow_select::window::shift (window.rs:56)
inline ArrayRef shift(const ArrayRef& values, int64_t offset) {
  ah_array_out out;
  values->context()->check(ah_shift(values->context()->handle(), &values->view(), offset, &out));
  return wrap(values, out, {values});
}
inline ArrayRef sort(const ArrayRef& values, SortOptions options = {}) { return take(values, sort_to_indices(values, options)); }
inline ArrayRef sort_limit(const ArrayRef& values, SortOptions options, int64_t limit) {
  return take(values, sort_to_indices(values, options, limit));
}

}  // namespace compute

// parquet RowSelection on device bitmaps (parquet/src/arrow/arrow_reader/selection/mod.rs): masks are Boolean arrays
// without nulls
namespace selection {
inline ArrayRef and_then(const ArrayRef& mask, const ArrayRef& other) {
  ah_array_out out;
  mask->context()->check(ah_selection_and_then(mask->context()->handle(), &mask->view(), &other->view(), &out));
  return compute::wrap(mask, out, {mask});
}
inline ArrayRef intersection(const ArrayRef& l, const ArrayRef& r) {
  ah_array_out out;
  l->context()->check(ah_selection_combine(l->context()->handle(), 0, &l->view(), &r->view(), &out));
  return compute::wrap(l, out);
}
inline ArrayRef union_(const ArrayRef& l, const ArrayRef& r) {
  ah_array_out out;
  l->context()->check(ah_selection_combine(l->context()->handle(), 1, &l->view(), &r->view(), &out));
  return compute::wrap(l, out);
}
inline ArrayRef boundaries(const ArrayRef& mask) {  // run starts (Int64): the RLE form
  ah_array_out out;
  mask->context()->check(ah_selection_boundaries(mask->context()->handle(), &mask->view(), &out));
  return compute::wrap(mask, out);
}
inline int64_t find_nth_set_bit(const ArrayRef& mask, int64_t start, int64_t n) {
  int64_t pos = 0;
  mask->context()->check(ah_selection_find_nth_set_bit(mask->context()->handle(), &mask->view(), start, n, &pos));
  return pos;
}
}  // namespace selection

// BatchCoalescer (arrow-select/src/coalesce.rs:148): exact-size output batches from a stream of (filtered) input
// batches.  The state machine is the native ah_coalescer object; primitive, Boolean and Utf8 / LargeUtf8 columns.
class BatchCoalescer {
 public:
  BatchCoalescer(std::shared_ptr<Context> ctx, const std::vector<ah_type>& types, int64_t target_batch_size)
      : ctx_(std::move(ctx)), n_(types.size()) {
    ctx_->check(ah_coalescer_create(ctx_->handle(), (int32_t)types.size(), types.data(), target_batch_size, &h_));
  }
  ~BatchCoalescer() { ah_coalescer_destroy(ctx_->handle(), h_); }
  BatchCoalescer(const BatchCoalescer&) = delete;
  BatchCoalescer& operator=(const BatchCoalescer&) = delete;
  BatchCoalescer& with_biggest_coalesce_batch_size(std::optional<int64_t> limit) {  // coalesce.rs:196
    ah_coalescer_set_biggest_coalesce_batch_size(h_, limit ? *limit : -1);
    return *this;
  }
  void push_batch(const RecordBatch& b) { push(b, nullptr); }                                       // :296
  void push_batch_with_filter(const RecordBatch& b, const ArrayRef& filter) { push(b, &filter); }   // :229
  // take_record_batch(batch, indices) into the in-progress batch (:289)
  void push_batch_with_indices(const RecordBatch& b, const ArrayRef& indices) {
    std::vector<ah_array_view> views;
    for (auto& c : b.columns) views.push_back(c->view());
    ctx_->check(ah_coalescer_push_batch_with_indices(ctx_->handle(), h_, views.data(), b.num_rows, &indices->view()));
  }
  // several filtered pushes handed over together: the same output as one push_batch_with_filter each, but ONE count
  // read-back for the group and the batches of one output window scattered by one launch (for hosts with batches queued)
  void push_batches_with_filters(const std::vector<std::pair<RecordBatch, ArrayRef>>& batches) {
    std::vector<ah_array_view> views, filters;
    std::vector<int64_t> rows;
    std::vector<uint64_t> tags;
    for (auto& bf : batches) {
      for (auto& c : bf.first.columns) views.push_back(c->view());
      filters.push_back(bf.second->view());
      rows.push_back(bf.first.num_rows);
      tags.push_back(next_tag_++);
    }
    std::vector<int32_t> bypass(batches.size(), 0);
    ctx_->check(ah_coalescer_push_batches_with_filters(ctx_->handle(), h_, (int32_t)batches.size(), views.data(), rows.data(),
                                                       filters.data(), tags.data(), bypass.data()));
    for (size_t i = 0; i < batches.size(); ++i)
      if (bypass[i]) bypassed_.emplace(tags[i], batches[i].first);
  }
  void finish_buffered_batch() { ctx_->check(ah_coalescer_finish_buffered_batch(ctx_->handle(), h_)); }  // :547; does not wait
  bool has_completed_batch() const { return ah_coalescer_completed_count(h_) > 0; }
  int64_t get_buffered_rows() const { return ah_coalescer_buffered_rows(h_); }
  bool is_empty() const { return get_buffered_rows() == 0 && !has_completed_batch(); }
  std::optional<RecordBatch> next_completed_batch() {  // :566
    std::vector<ah_array_out> outs(n_);
    int64_t rows = -1;
    uint64_t tag = 0;
    ctx_->check(ah_coalescer_next_completed_batch(ctx_->handle(), h_, outs.data(), &rows, &tag));
    if (rows < 0) return std::nullopt;
    if (tag) {  // the caller's own batch, passed through untouched (large-batch bypass)
      RecordBatch b = std::move(bypassed_.at(tag));
      bypassed_.erase(tag);
      return b;
    }
    RecordBatch b;
    b.num_rows = rows;
    for (size_t i = 0; i < n_; ++i) b.columns.push_back(std::make_shared<Array>(ctx_, outs[i]));
    return b;
  }

 private:
  void push(const RecordBatch& b, const ArrayRef* filter) {
    std::vector<ah_array_view> views;
    for (auto& c : b.columns) views.push_back(c->view());
    const uint64_t tag = next_tag_++;
    int32_t bypass = 0;
    if (filter)
      ctx_->check(ah_coalescer_push_batch_with_filter(ctx_->handle(), h_, views.data(), b.num_rows, &(*filter)->view(), tag, &bypass));
    else
      ctx_->check(ah_coalescer_push_batch(ctx_->handle(), h_, views.data(), b.num_rows, tag, &bypass));
    if (bypass) bypassed_.emplace(tag, b);
  }
  std::shared_ptr<Context> ctx_;
  ah_coalescer* h_ = nullptr;
  size_t n_;
  uint64_t next_tag_ = 1;
  std::map<uint64_t, RecordBatch> bypassed_;
};

// The multi-GPU exchange: one process per GPU, one Communicator per process (ah_comm_*; libarrow_hip.so binds RCCL).
// Rank 0 creates the id, the host ships its 128 bytes to every rank, every rank constructs a Communicator with it.
class Communicator {
 public:
  static std::array<uint8_t, AH_COMM_ID_BYTES> unique_id(const std::shared_ptr<Context>& ctx) {
    std::array<uint8_t, AH_COMM_ID_BYTES> id{};
    ctx->check(ah_comm_unique_id(ctx->handle(), id.data()));
    return id;
  }
  Communicator(std::shared_ptr<Context> ctx, int rank, int world, const uint8_t* id) : ctx_(std::move(ctx)) {
    ctx_->check(ah_comm_create(ctx_->handle(), rank, world, id, &h_));
  }
  ~Communicator() { ah_comm_destroy(ctx_->handle(), h_); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  // concat(rank 0's array, rank 1's array, ...) on every rank (arrow-select/src/concat.rs:495 across ranks)
  ArrayRef all_gatherv(const ArrayRef& local, ah_exchange_stats* stats = nullptr) {
    ah_array_out out;
    ctx_->check(ah_all_gatherv(ctx_->handle(), h_, &local->view(), &out, stats));
    return std::make_shared<Array>(ctx_, out);
  }
  // concat_batches (concat.rs:607) of every rank's shard: one count exchange + one grouped send / recv for all columns
  RecordBatch all_gather_record_batch(const RecordBatch& shard, ah_exchange_stats* stats = nullptr) {
    std::vector<ah_array_view> views;
    for (auto& c : shard.columns) views.push_back(c->view());
    std::vector<ah_array_out> outs(views.size());
    ctx_->check(ah_all_gather_columns(ctx_->handle(), h_, (int32_t)views.size(), views.data(), outs.data(), stats));
    RecordBatch b;
    for (auto& o : outs) b.columns.push_back(std::make_shared<Array>(ctx_, o));
    b.num_rows = b.columns.empty() ? 0 : b.columns[0]->len();
    return b;
  }
  // The same exchange split in two: `begin` enqueues the count exchange, the sends / receives and the merges and returns;
  // independent work (a take on another context, the next filter) runs meanwhile; `end()` waits and hands the batch over.
  // The shard's buffers must stay alive until end() (the handle holds them).
  class Pending {
   public:
    Pending(Communicator* c, ah_exchange* x, RecordBatch shard) : c_(c), x_(x), shard_(std::move(shard)) {}
    Pending(Pending&& o) noexcept : c_(o.c_), x_(o.x_), shard_(std::move(o.shard_)) { o.x_ = nullptr; }
    Pending(const Pending&) = delete;
    ~Pending() {  // abandoned: finish it anyway (peers are in the collective) and drop the result
      if (!x_) return;
      std::vector<ah_array_out> outs(shard_.columns.size());
      if (ah_all_gather_columns_end(c_->ctx_->handle(), c_->h_, x_, outs.data(), nullptr) == AH_OK)
        for (auto& o : outs) ah_array_release(c_->ctx_->handle(), &o);
    }
    RecordBatch end(ah_exchange_stats* stats = nullptr) {
      std::vector<ah_array_out> outs(shard_.columns.size());
      ah_exchange* x = x_;
      x_ = nullptr;
      c_->ctx_->check(ah_all_gather_columns_end(c_->ctx_->handle(), c_->h_, x, outs.data(), stats));
      RecordBatch b;
      for (auto& o : outs) b.columns.push_back(std::make_shared<Array>(c_->ctx_, o));
      b.num_rows = b.columns.empty() ? 0 : b.columns[0]->len();
      return b;
    }

   private:
    Communicator* c_;
    ah_exchange* x_;
    RecordBatch shard_;
  };
  Pending all_gather_record_batch_begin(const RecordBatch& shard) {
    std::vector<ah_array_view> views;
    for (auto& c : shard.columns) views.push_back(c->view());
    ah_exchange* x = nullptr;
    ctx_->check(ah_all_gather_columns_begin(ctx_->handle(), h_, (int32_t)views.size(), views.data(), &x));
    return Pending(this, x, shard);
  }
  void barrier() { ctx_->check(ah_comm_barrier(ctx_->handle(), h_)); }
  // element-wise MAX over ranks (the bench's "slowest rank" clock), in place
  void allreduce_max(std::vector<double>& values) {
    ctx_->check(ah_comm_allreduce_max_f64(ctx_->handle(), h_, values.data(), (int32_t)values.size()));
  }

 private:
  std::shared_ptr<Context> ctx_;
  ah_comm* h_ = nullptr;
};

// Arrow C Data Interface (arrow-array/src/ffi.rs:231-254).  `from_ffi` copies a host-resident
// producer array into HBM (the producer keeps ownership of its structs); `to_ffi` fills structs
// whose release callbacks free host copies of the device buffers.
namespace ffi {
inline ArrayRef from_ffi(const std::shared_ptr<Context>& ctx, const ArrowArray& array, const ArrowSchema& schema) {
  ah_array_out out;
  ctx->check(ah_import_c_data(ctx->handle(), &array, &schema, &out));
  return std::make_shared<Array>(ctx, out);
}
// `format` keeps the logical type ("tsu:UTC", "d:38,10"); nullptr = default of the physical type
inline void to_ffi(const ArrayRef& a, ArrowArray* out_array, ArrowSchema* out_schema, const char* format = nullptr) {
  a->context()->check(ah_export_c_data(a->context()->handle(), &a->view(), format, out_array, out_schema));
}
}  // namespace ffi
}  // namespace arrow_hip
