/*
 * oracle.h — CPU oracle for the arrow-rs compute hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, single-threaded C++ restatement
 * of the reference's algorithms (apache/arrow-rs 59.2.0) — same pass structure,
 * same edge semantics — used as (a) the parity checker for the HIP path in
 * tests/, __graft_entry__.smoke() and (b) bench.py's `cpu_baseline` leg
 * (kind "port").  Nothing under arrow-rs_amd/ may include, link or call it.
 *
 * Parity pinning: the reference is Rust-only and there is no Rust toolchain in
 * this image, so oracle/_ref cannot be built.  The oracle is pinned against
 * every inline golden vector the reference's own tests hold for this path
 * (tests/golden/ *.json, transcribed from the files cited there) and
 * cross-checked against pyarrow (Arrow C++) where semantics coincide.
 * Float64->Utf8 follows the published Ryu algorithm (ryu crate 1.0.23, not
 * vendored under /root/reference): beyond the five values the reference's tests
 * pin, that arm is "parity unpinned" by the reference itself; it is
 * additionally checked against CPython's repr() shortest-digit output.
 *
 * Struct layouts intentionally equal include/arrow_hip.h's ah_array_view /
 * ah_array_out so one ctypes definition serves both sides; pointers here are
 * HOST pointers.
 */
#ifndef ARROW_ORACLE_H
#define ARROW_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as AH_* */
enum {
  ORC_OK = 0, ORC_INVALID_ARGUMENT = 1, ORC_COMPUTE_ERROR = 2, ORC_ARITHMETIC_OVERFLOW = 3,
  ORC_DIVIDE_BY_ZERO = 4, ORC_CAST_ERROR = 5, ORC_OFFSET_OVERFLOW = 6, ORC_NOT_YET_IMPLEMENTED = 7,
  ORC_OFFSET_OVERFLOW_ERROR = 8,
  ORC_PANIC = 100
};
enum {
  ORC_BOOL = 1, ORC_INT8 = 2, ORC_INT16 = 3, ORC_INT32 = 4, ORC_INT64 = 5, ORC_UINT8 = 6,
  ORC_UINT16 = 7, ORC_UINT32 = 8, ORC_UINT64 = 9, ORC_FLOAT32 = 10, ORC_FLOAT64 = 11,
  ORC_FIXED16 = 12, ORC_FIXED32 = 13, ORC_UTF8 = 14, ORC_LARGE_UTF8 = 15, ORC_FLOAT16 = 16
};

typedef struct orc_view {
  int32_t type;
  int64_t length;
  int64_t null_count; /* -1: count it */
  const void* values;
  int64_t values_bit_offset;
  const uint8_t* validity;
  int64_t validity_bit_offset;
  const void* offsets; /* ORC_UTF8 / ORC_LARGE_UTF8: length+1 offsets; values = byte data */
} orc_view;

typedef struct orc_out {
  int32_t type;
  int64_t length;
  int64_t null_count;
  void* values;
  int64_t values_bytes;
  int64_t values_bit_offset;
  uint8_t* validity;
  int64_t validity_bytes;
  int64_t validity_bit_offset;
  void* offsets;
  int64_t offsets_bytes;
  int32_t flags; /* 1 = borrowed (zero-copy slice of the input) */
} orc_out;

const char* orc_last_error(void);
void orc_release(orc_out* out);

int64_t orc_count_set_bits(const uint8_t* bits, int64_t bit_offset, int64_t len);
/* BitSliceIterator / BitIndexIterator: write up to cap entries, return the count */
int64_t orc_set_slices(const uint8_t* bits, int64_t bit_offset, int64_t len, int64_t* out_pairs,
                       int64_t cap);
int64_t orc_set_indices(const uint8_t* bits, int64_t bit_offset, int64_t len, int64_t* out_idx,
                        int64_t cap);

int32_t orc_filter(const orc_view* values, const orc_view* predicate, orc_out* out);
/* IterationStrategy::default_strategy (filter.rs:346-364): 0 None, 1 All, 2 SlicesIterator, 3 IndexIterator */
int32_t orc_filter_strategy(const orc_view* predicate);
int32_t orc_take(const orc_view* values, const orc_view* indices, int32_t check_bounds, orc_out* out);
int32_t orc_arith(int32_t op, const orc_view* lhs, int32_t lhs_scalar, const orc_view* rhs,
                  int32_t rhs_scalar, orc_out* out);
int32_t orc_neg(const orc_view* values, int32_t wrapping, orc_out* out);
int32_t orc_compare(int32_t op, const orc_view* lhs, int32_t lhs_scalar, const orc_view* rhs,
                    int32_t rhs_scalar, orc_out* out);
/* arrow_arith::boolean: op numbering as AH_BOOL_* */
int32_t orc_boolean_binary(int32_t op, const orc_view* l, const orc_view* r, orc_out* out);
int32_t orc_boolean_unary(int32_t op, const orc_view* v, orc_out* out);
/* arrow_string::like::{like,nlike,starts_with,ends_with,contains} with a scalar pattern (arrow-string/src/like.rs:83-205,
 * predicate.rs:44-306); op: 0 like, 1 nlike, 2 starts_with, 3 ends_with, 4 contains. */
int32_t orc_string_like(int32_t op, const orc_view* values, const orc_view* pattern, orc_out* out);
/* arrow_string::length::{length, bit_length} (arrow-string/src/length.rs:58,:130) */
int32_t orc_string_length(const orc_view* values, int32_t bits, orc_out* out);
int32_t orc_nullif(const orc_view* left, const orc_view* right, orc_out* out);
int32_t orc_cast(const orc_view* values, int32_t to_type, int32_t safe, orc_out* out);
/* cast_with_options for pairs where either side is a temporal logical type (arrow-cast/src/cast/mod.rs:1700-2260);
 * the descriptor has include/arrow_hip.h's ah_data_type layout (id: ORC_* or 32 Date32, 33 Date64, 34 Time32,
 * 35 Time64, 36 Timestamp, 37 Duration; unit 0 s, 1 ms, 2 us, 3 ns; zones as fixed offsets). */
typedef struct orc_data_type {
  int32_t id;
  int32_t unit;
  int32_t has_tz;
  int32_t tz_offset_seconds;
  int32_t precision;
  int32_t scale;
} orc_data_type;
int32_t orc_cast_with_types(const orc_view* values, const orc_data_type* from, const orc_data_type* to, int32_t safe,
                            orc_out* out);
/* arithmetic_op's type rules for temporal operands (arrow-arith/src/numeric.rs:225-275, :426-537, :877-932): op as
 * AH_ADD..AH_REM; the result's logical type is written to out_type. */
int32_t orc_arith_with_types(int32_t op, const orc_view* lhs, int32_t lhs_scalar, const orc_data_type* lhs_type,
                             const orc_view* rhs, int32_t rhs_scalar, const orc_data_type* rhs_type, orc_out* out,
                             orc_data_type* out_type);
int32_t orc_concat(int32_t n, const orc_view* pieces, orc_out* out);

/* arrow_arith::aggregate (arrow-arith/src/aggregate.rs): op numbering as AH_AGG_*.
 * `vector_bytes` is the reference's compile-time PREFERRED_VECTOR_SIZE (:300-307): 16 for the
 * default x86_64 target, 32 with AVX, 64 with AVX-512; 0 = 16.  It only changes the association
 * order of FLOAT sums/products (every other result is order-independent). */
typedef struct orc_scalar {
  int32_t type;
  int32_t is_valid; /* 0 = None */
  uint8_t bytes[32];
} orc_scalar;
int32_t orc_aggregate(int32_t op, const orc_view* values, int32_t vector_bytes, orc_scalar* out);

/* parquet RowSelection, bitmap-backed form (parquet/src/arrow/arrow_reader/selection/): masks are ORC_BOOL views
 * without nulls.  and_then: algebra.rs:392-436 (panic texts as ORC_PANIC); combine op 0 = intersect_masks,
 * 1 = union_masks (:267-349, the longer side's tail passes through); find_nth_set_bit_position:
 * arrow-buffer/src/buffer/boolean.rs:445-455. */
int32_t orc_selection_and_then(const orc_view* mask, const orc_view* other, orc_out* out);
int32_t orc_selection_combine(int32_t op, const orc_view* l, const orc_view* r, orc_out* out);
int64_t orc_find_nth_set_bit(const uint8_t* bits, int64_t bit_offset, int64_t len, int64_t start, int64_t n);

/* arrow_ord::sort::sort_to_indices (arrow-ord/src/sort.rs:276): UInt32 indices; limit < 0 = None.  The
 * reference sorts with `sort_unstable_by`, so the order of equal keys is unspecified there; this restatement
 * (and the device) use the stable order — equal keys stay in ascending index order, under the reversed comparator
 * too — which is also what every tie in the reference's own tests shows (sort.rs:1625-1895). */
int32_t orc_sort_to_indices(const orc_view* values, int32_t descending, int32_t nulls_first, int64_t limit,
                            orc_out* out);

/* lexsort_to_indices (sort.rs:939) with `make_comparator` semantics per column (null == null, null before / after
 * valid by nulls_first, descending reverses valid comparisons); stable, so fully equal rows keep row order. */
int32_t orc_rank(const orc_view* a, int32_t desc, int32_t nulls_first, orc_out* out);
int32_t orc_lexsort_to_indices(int32_t n_cols, const orc_view* cols, const int32_t* descending,
                               const int32_t* nulls_first, int64_t limit, orc_out* out);

/* arrow_select::zip::zip (arrow-select/src/zip.rs:99): out[i] = mask[i] (null = false) ? truthy : falsy; either side may
 * be a length-1 scalar.  Fixed-width and Boolean layouts. */
int32_t orc_zip(const orc_view* mask, const orc_view* truthy, int32_t truthy_scalar, const orc_view* falsy,
                int32_t falsy_scalar, orc_out* out);

/* arrow_select::interleave::interleave (arrow-select/src/interleave.rs:74) for fixed-width / Boolean arrays;
 * indices as two UInt32 arrays (array, row). */
int32_t orc_interleave(int32_t n, const orc_view* arrays, const uint32_t* array_index, const uint32_t* row_index,
                       int64_t n_indices, orc_out* out);

/* arrow_arith::bitwise (arrow-arith/src/bitwise.rs:42-205): op 8 and, 9 or, 10 xor, 11 shift_left, 12 shift_right,
 * 13 and_not (numbering as AH_BIT_*), 14 not (unary: rhs ignored); integers only; `binary` / `unary` null rules. */
int32_t orc_bitwise(int32_t op, const orc_view* lhs, int32_t lhs_scalar, const orc_view* rhs, int32_t rhs_scalar,
                    orc_out* out);

/* format one f64/f32 the way ryu::Buffer::format does; returns the length */
int32_t orc_format_f64(double v, char* buf /* >= 32 */);
int32_t orc_format_f32(float v, char* buf /* >= 32 */);

/* host twins of the device generators in arrow-rs_amd/csrc/gen.hip */
void orc_gen_uniform_i64(int64_t* dst, int64_t n, uint64_t seed, int64_t lo, int64_t hi, int64_t row0);
void orc_gen_uniform_i32(int32_t* dst, int64_t n, uint64_t seed, int64_t row0);
void orc_gen_uniform_u32(uint32_t* dst, int64_t n, uint64_t seed, uint32_t bound, int64_t row0);
void orc_gen_uniform_f64(double* dst, int64_t n, uint64_t seed, double lo, double hi, int64_t row0);
void orc_gen_bernoulli_bits(uint8_t* dst, int64_t n, uint64_t seed, double p_true, int64_t row0);
void orc_zero_null_slots(void* values, int32_t byte_width, const uint8_t* validity, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
