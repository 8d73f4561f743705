/*
 * arrow_hip.h — C ABI of the MI355X-native columnar compute hot path.
 *
 * This is the drop-in boundary behind arrow-rs's `arrow::compute::kernels`
 * function shapes (reference: arrow/src/compute/kernels.rs:20-27).  The
 * reference has no FFI boundary around its kernels; the two ABI-stable
 * surfaces it does define are foreign buffer ownership
 * (arrow-buffer/src/buffer/immutable.rs:170-176 `Buffer::from_custom_allocation`)
 * and the Arrow C Data Interface (arrow-data/src/ffi.rs:37-66).  A Rust shim
 *  wraps device buffers returned from here with
 * `from_custom_allocation` and never reads device bytes on the host: every
 * entry point therefore returns `length` and `null_count` explicitly.
 *
 * Conventions
 *  - all `values` / `validity` / `offsets` pointers are DEVICE pointers (HBM);
 *  - bitmaps are LSB-first bit-packed with an arbitrary bit offset
 *    (arrow-buffer/src/buffer/boolean.rs:97-104); validity == NULL means
 *    "no nulls"; 1 = valid;
 *  - inputs are borrowed for the duration of the call; outputs are freshly
 *    allocated through the context allocator and owned by the caller until
 *    `ah_array_release` (exception: AH_OUT_BORROWED zero-copy fast paths,
 *    reference filter.rs:546 `values.slice(0, count)`);
 *  - every entry point is synchronous at return (results usable immediately),
 *    re-entrant across contexts; one context must not be used from two threads
 *    at once (the reference kernels are pure functions over Send+Sync arrays —
 *    use one context per calling thread);
 *  - errors: integer status mirroring `ArrowError` variants
 *    (arrow-schema/src/error.rs:26-69) + `ah_last_error()` reproducing the
 *    reference's message text; reference *panics* map to AH_PANIC.
 */
#ifndef ARROW_HIP_H
#define ARROW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AH_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- status */
typedef int32_t ah_status;
enum {
  AH_OK = 0,
  AH_INVALID_ARGUMENT = 1,    /* ArrowError::InvalidArgumentError */
  AH_COMPUTE_ERROR = 2,       /* ArrowError::ComputeError */
  AH_ARITHMETIC_OVERFLOW = 3, /* ArrowError::ArithmeticOverflow */
  AH_DIVIDE_BY_ZERO = 4,      /* ArrowError::DivideByZero */
  AH_CAST_ERROR = 5,          /* ArrowError::CastError */
  AH_OFFSET_OVERFLOW = 6,     /* "byte array offset overflow" (a panic in the ref,
                                 arrow-array/src/builder/generic_bytes_builder.rs:86-87) */
  AH_NOT_YET_IMPLEMENTED = 7, /* ArrowError::NotYetImplemented */
  AH_OFFSET_OVERFLOW_ERROR = 8, /* ArrowError::OffsetOverflowError(n): message is the number */
  AH_C_DATA_INTERFACE = 9,    /* ArrowError::CDataInterface */
  AH_IPC_ERROR = 10,          /* ArrowError::IpcError */
  AH_PARSE_ERROR = 11,        /* ArrowError::ParseError */
  AH_PANIC = 100,             /* the reference would panic!(); message = panic text */
  AH_HIP_ERROR = 101,         /* runtime failure (no reference analogue) */
  AH_OUT_OF_MEMORY = 102,
  AH_COMM_ERROR = 103         /* RCCL failure in the multi-GPU exchange (no reference analogue) */
};

/* ----------------------------------------------------------------- types */
/* Physical layouts only: logical types that share a layout (Date32, Timestamp,
 * Duration, Decimal128...) are preserved by the host wrapper exactly as the
 * reference preserves `data_type` (filter.rs:783-787, take.rs:414). */
typedef int32_t ah_type;
enum {
  AH_BOOL = 1, /* bit-packed values */
  AH_INT8 = 2, AH_INT16 = 3, AH_INT32 = 4, AH_INT64 = 5,
  AH_UINT8 = 6, AH_UINT16 = 7, AH_UINT32 = 8, AH_UINT64 = 9,
  AH_FLOAT32 = 10, AH_FLOAT64 = 11,
  AH_FIXED16 = 12, /* 16-byte natives: i128 / Decimal128 / IntervalMonthDayNano */
  AH_FIXED32 = 13, /* 32-byte natives: i256 / Decimal256 */
  AH_UTF8 = 14,       /* i32 offsets: output of cast; input/output of filter, take */
  AH_LARGE_UTF8 = 15, /* i64 offsets */
  AH_FLOAT16 = 16,    /* half::f16: filter/take/concat (bit copy), sort, min/max; arithmetic, neg, compare and the numeric
                       * casts computed as the `half` crate does (to f32, one operation, one rounding back; totalOrder on
                       * the 16-bit pattern) — numeric.rs:113,240, cast/mod.rs:1578-1697 */
  /* GenericByteViewArray (arrow-array/src/array/byte_view_array.rs): `values` = the 16-byte views.
   * filter / take / nullif copy views unchanged and the host keeps the SAME variadic data-buffer list
   * on the result (filter.rs:931-944 `filter_byte_view`, take.rs:630-640 `take_byte_view`), so the data
   * buffers never enter the C ABI.  concat is refused (it would have to renumber buffer indices). */
  AH_UTF8_VIEW = 17, AH_BINARY_VIEW = 18
};

/* Borrowed view of a PrimitiveArray / BooleanArray living in HBM
 * (reference: arrow-array/src/array/primitive_array.rs:596-601,
 *  arrow-array/src/array/boolean_array.rs:68). */
typedef struct ah_array_view {
  ah_type type;
  int64_t length;              /* rows */
  int64_t null_count;          /* >=0 known; -1 = unknown, the library counts */
  const void* values;          /* primitives: pointer ALREADY advanced by the slice
                                  offset (ScalarBuffer semantics, scalar.rs:188-209),
                                  element-aligned.  AH_BOOL: bit-packed base. */
  int64_t values_bit_offset;   /* AH_BOOL only */
  const uint8_t* validity;     /* NULL = no nulls */
  int64_t validity_bit_offset;
  const void* offsets;         /* AH_UTF8 (i32) / AH_LARGE_UTF8 (i64) only: length+1 offsets,
                                  pointer already advanced by the slice offset; `values` = byte data */
} ah_array_view;

enum {
  AH_OUT_BORROWED = 1,        /* every buffer aliases the input (zero-copy slice) */
  AH_OUT_BORROWED_VALUES = 2  /* values/offsets alias the input, validity is owned (nullif) */
};

/* Owned result.  Freshly produced bitmaps start at bit offset 0. */
typedef struct ah_array_out {
  ah_type type;
  int64_t length;
  int64_t null_count;
  void* values;            /* primitives: values; AH_BOOL: bit-packed; UTF8: bytes */
  int64_t values_bytes;    /* size of the ALLOCATION behind `values` (what the allocator hook's free receives); the array
                              uses the first length * width bytes of it — filter of a small batch allocates for the
                              worst case (every row selected) so that it need not wait for the count first */
  int64_t values_bit_offset;   /* AH_BOOL only; 0 unless AH_OUT_BORROWED */
  uint8_t* validity;       /* NULL when the result carries no null buffer */
  int64_t validity_bytes;
  int64_t validity_bit_offset; /* 0 unless AH_OUT_BORROWED */
  void* offsets;           /* AH_UTF8 / AH_LARGE_UTF8: length+1 offsets.  AH_UTF8_VIEW PRODUCED by ah_cast: the result's one
                              variadic data buffer (buffer index 0 of its views: the strings longer than 12 bytes, back
                              to back), NULL when every string is inline; views that pass through filter / take keep
                              referring to the caller's buffers and leave this NULL */
  int64_t offsets_bytes;
  int32_t flags;
} ah_array_out;

/* Logical types.  The casts and the arithmetic whose result depends on the LOGICAL type (arrow-cast/src/cast/mod.rs:
 * 1700-2260, the "temporal casts"; arrow-arith/src/numeric.rs:426-932) cannot work from the physical ah_type alone —
 * it cannot tell Timestamp(Second) from Timestamp(Millisecond) — so ah_cast_with_types / ah_arith_with_types take the
 * reference's DataType as a small descriptor.  `id` is an ah_type for the plain types (AH_INT32, AH_FLOAT64, ...,
 * the other fields 0) or one of AH_DT_*; `unit` is the TimeUnit of Time32 / Time64 / Timestamp / Duration; a
 * Timestamp's timezone is carried as a FIXED UTC offset in seconds (`has_tz`, `tz_offset_seconds`: "+05:45" = 20700 —
 * the only zones the reference parses without its optional chrono-tz feature, arrow-array/src/timezone.rs); a host
 * with a zone database resolves named zones itself.  `precision` / `scale` are reserved for the decimal arms. */
typedef int32_t ah_time_unit;
enum { AH_SECOND = 0, AH_MILLISECOND = 1, AH_MICROSECOND = 2, AH_NANOSECOND = 3 };
enum {
  AH_DT_DATE32 = 32,    /* i32 days since the epoch */
  AH_DT_DATE64 = 33,    /* i64 milliseconds since the epoch */
  AH_DT_TIME32 = 34,    /* i32, unit = second | millisecond */
  AH_DT_TIME64 = 35,    /* i64, unit = microsecond | nanosecond */
  AH_DT_TIMESTAMP = 36, /* i64 since the epoch in `unit`, optional fixed-offset zone */
  AH_DT_DURATION = 37,  /* i64 in `unit` */
  AH_DT_DECIMAL128 = 39, /* i128 (AH_FIXED16) with `precision` / `scale` */
  AH_DT_INTERVAL = 38   /* unit = 0 YearMonth (i32) | 1 DayTime (8 bytes) | 2 MonthDayNano (16 bytes); named only so that
                           ah_arith_with_types can refuse it with the reference's type text: no kernel takes intervals yet */
};
typedef struct ah_data_type {
  int32_t id;
  ah_time_unit unit;
  int32_t has_tz;
  int32_t tz_offset_seconds;
  int32_t precision;
  int32_t scale;
} ah_data_type;

/* --------------------------------------------------------------- context */
typedef struct ah_context ah_context;
typedef void* (*ah_alloc_fn)(void* user, size_t bytes); /* returns device ptr, 256B aligned */
typedef void (*ah_free_fn)(void* user, void* ptr, size_t bytes);

/* GPUs visible to this process (hipGetDeviceCount); 0 when there is none or the runtime cannot start. */
AH_API int32_t ah_device_count(void);
AH_API ah_status ah_context_create(int device, ah_context** out);
AH_API void ah_context_destroy(ah_context* ctx);
/* Route OUTPUT allocations through the host's allocator (the hook a Rust host
 * uses to own results as `Buffer::from_custom_allocation`); NULL restores the
 * built-in pooled hipMalloc allocator. */
AH_API void ah_context_set_allocator(ah_context* ctx, ah_alloc_fn a, ah_free_fn f, void* user);
/* Launch on a caller-provided hipStream_t (NULL = the context's own stream). */
AH_API void ah_context_set_stream(ah_context* ctx, void* hip_stream);
AH_API void* ah_context_stream(ah_context* ctx);
/* Opt-in asynchronous ("deferred") mode.  The reference's kernels are synchronous functions; ours are too
 * by default: every entry point returns with its result complete and `null_count` known.  With deferred mode
 * on, the entry points whose output SHAPE does not depend on the data and that cannot fail on the device —
 * ah_arith_binary / ah_arith_neg for the wrapping, floating-point and bitwise ops, ah_bitwise_not, ah_compare,
 * ah_boolean_binary / ah_boolean_unary, ah_cast between numeric types in safe mode,
 * ah_filter_predicate_apply on fixed-width and Boolean values (the row count comes from the predicate), and ah_take of
 * fixed-width / Boolean values without check_bounds (indices with a known null count or no validity) — only
 * ENQUEUE their kernels on the context's stream and return at once: no host synchronisation, `null_count = -1`
 * ("unknown", the C Data Interface's convention; every entry point accepts -1 on its inputs) and the validity
 * buffer kept even where the synchronous call would have dropped an all-valid one.  Results may be passed
 * straight to further calls on the same context (stream order).  Before reading them from the host, another
 * stream or another context call ah_synchronize(); ah_array_resolve() does that and fills in null_count.
 * A deferred ah_take that meets an out-of-bounds index cannot report it at return: the reference's panic (same AH_PANIC
 * text as the synchronous call, take.rs:447,454) is returned by the NEXT ah_synchronize / ah_array_resolve on the context —
 * the first such fault in stream order; the take's own output is then garbage at that row, as after a caught panic.
 * Every other entry point (data-dependent sizes, device-side errors to report) stays synchronous. */
AH_API void ah_context_set_deferred(ah_context* ctx, int32_t on);
AH_API int32_t ah_context_deferred(const ah_context* ctx);
/* hipGraph capture of deferred calls — the launch-bound regime (query-engine batches of 10^3 .. 10^5 rows: 2-4 us of
 * kernel behind 3-4 us of host launch cost per kernel) as ONE graph launch.  Between ah_graph_begin and ah_graph_end
 * the entry points deferred mode covers (see above) are RECORDED on the context's stream, not run: they return their
 * ah_array_out as usual (null_count = -1) but the buffers are written only when the graph is launched.  ah_graph_launch
 * enqueues one replay of the whole sequence: it reads whatever bytes the captured INPUT pointers hold at that moment
 * and rewrites the SAME output buffers the captured calls returned — keep inputs, outputs and (for filters) the
 * prebuilt ah_filter_predicate alive and unmoved for as long as the graph lives; shapes (lengths, the predicate and
 * hence K) are frozen at capture.  An entry point that has to wait on the device (data-dependent sizes, checked
 * arithmetic, take, strings, ah_synchronize) fails fast while recording and invalidates the capture (ah_graph_end then
 * returns the error).  Scratch released while recording stays reserved until ah_graph_destroy. */
typedef struct ah_graph ah_graph;
AH_API ah_status ah_graph_begin(ah_context* ctx);
AH_API ah_status ah_graph_end(ah_context* ctx, ah_graph** out);
AH_API int32_t ah_graph_node_count(const ah_graph* g);
AH_API ah_status ah_graph_launch(ah_context* ctx, ah_graph* g);
AH_API void ah_graph_destroy(ah_context* ctx, ah_graph* g);
/* ah_synchronize + count the nulls of a deferred result (null_count < 0) on the device. */
AH_API ah_status ah_array_resolve(ah_context* ctx, ah_array_out* out);
/* Memory accounting of a context's built-in pooled allocator — the reference's MemoryPool::used() / TrackingMemoryPool
 * (arrow-buffer/src/pool.rs:73-93) for device memory: what a host samples to size its batches and what bench.py /
 * a profiler reads next to HBM traffic.  Sizes are the pool's ROUNDED block sizes (what is really held).  Outputs routed
 * through a host allocator hook (ah_context_set_allocator) are the host's to account for and do not appear here. */
typedef struct ah_context_stats_t {
  int64_t live_bytes;                /* held by blocks handed out and not yet released (results + scratch) */
  int64_t high_water_bytes;          /* maximum of live_bytes since creation / the last reset */
  int64_t cached_bytes;              /* released blocks kept on the free lists (ah_pool_trim returns them to HIP) */
  int64_t reserved_high_water_bytes; /* maximum of live + cached: what the context has had hipMalloc'ed at once */
  int64_t allocated_bytes_total;     /* cumulative */
  int64_t freed_bytes_total;
  int64_t alloc_calls;
  int64_t free_calls;
  int64_t pool_hits;                 /* allocations served from the free lists */
  int64_t device_malloc_calls;       /* allocations that went to hipMalloc */
  int64_t host_to_device_bytes;      /* cumulative payload bytes the host boundary moved over PCIe: ah_memcpy_htod and */
  int64_t device_to_host_bytes;      /* ah_import_c_data / ah_memcpy_dtoh and ah_export_c_data (kernel traffic stays in HBM) */
} ah_context_stats_t;
AH_API ah_status ah_context_stats(ah_context* ctx, ah_context_stats_t* out, int32_t reset_peaks);
AH_API const char* ah_last_error(ah_context* ctx);
AH_API void ah_array_release(ah_context* ctx, ah_array_out* out);
AH_API const char* ah_version(void);

/* raw device memory helpers for hosts without their own HIP bindings */
AH_API ah_status ah_device_alloc(ah_context* ctx, size_t bytes, void** out);
AH_API void ah_device_free(ah_context* ctx, void* ptr);
AH_API ah_status ah_memcpy_htod(ah_context* ctx, void* dst, const void* src, size_t bytes);
AH_API ah_status ah_memcpy_dtoh(ah_context* ctx, void* dst, const void* src, size_t bytes);
AH_API ah_status ah_memcpy_dtod(ah_context* ctx, void* dst, const void* src, size_t bytes);
AH_API ah_status ah_memset(ah_context* ctx, void* dst, int value, size_t bytes);
AH_API ah_status ah_synchronize(ah_context* ctx);
AH_API void ah_pool_trim(ah_context* ctx); /* hipFree everything cached in the pool */

/* ---------------------------------------------------------------- filter */
/* arrow_select::filter::filter (arrow-select/src/filter.rs:201).
 * predicate must be AH_BOOL; predicate nulls select nothing (filter.rs:167-171);
 * predicate.length <= values.length else AH_INVALID_ARGUMENT with the text of
 * filter.rs:537-541.  values may be any primitive type or AH_BOOL. */
AH_API ah_status ah_filter(ah_context* ctx, const ah_array_view* values,
                           const ah_array_view* predicate, ah_array_out* out);
/* Predicates of at most 2^20 rows over fixed-width columns (ah_filter and ah_filter_record_batch alike) take a
 * one-launch path: count, prefix and scatter in a single kernel, every column of a record batch in that launch, one host
 * wait for K and all null counts (query-engine batch sizes are latency, not bandwidth: arrow/benches/filter_kernels.rs:
 * 39-45 runs 512 .. 65 536 rows).  Same results as the general two-pass path; environment AH_FILTER_SMALL=0 disables it.
 * Larger predicates that select at most 1 row in 32 (the reference bench's "kept 1/1024" shapes) scatter through a
 * tile-per-wave kernel instead of the LDS-staged one (AH_FILTER_SPARSE=0 / 1 forces either); again the same results. */

/* FilterBuilder::new(..).optimize().build() (filter.rs:256-324): count once,
 * keep per-tile offsets on device, apply to many columns.  The handle BORROWS the
 * predicate's device buffers (the Rust FilterPredicate holds an Arc clone): they
 * must stay alive until ah_filter_predicate_free. */
typedef struct ah_filter_predicate ah_filter_predicate;
AH_API ah_status ah_filter_predicate_build(ah_context* ctx, const ah_array_view* predicate,
                                           ah_filter_predicate** out);
/* the counts of n (<= 128) predicates with ONE host wait (every count pass is enqueued first) */
AH_API ah_status ah_filter_predicates_build(ah_context* ctx, int32_t n, const ah_array_view* predicates,
                                            ah_filter_predicate** outs);
/* The LAZY form of FilterBuilder for predicates an engine computes on the fly — `filter(a, and_kleene(lt(a, x),
 * gt_eq(b, y)))` (arrow-ord/src/cmp.rs:113-164 compare_op :220-382; arrow-arith/src/boolean.rs:60-300;
 * filter.rs:201,256-273): instead of materialising every comparison and the boolean kernel's result, hand over the
 * TERMS.  term_0 join_0 term_1 join_1 ... is folded left to right; each term is a comparison (AH_EQ .. AH_GT_EQ) of
 * two Datums of one integer or Float32 / Float64 type, at least one of them an array; joins are AH_BOOL_AND, AH_BOOL_OR,
 * AH_BOOL_AND_KLEENE or AH_BOOL_OR_KLEENE.  The comparisons are evaluated inside the filter's count pass (__ballot
 * words, null propagation with the reference's bit formulas, null predicate rows select nothing) and the resulting
 * predicate is used like any other: ah_filter_predicate_count / _apply / _apply_into / the coalescer.  Result-identical
 * to the materialised chain; errors carry compare_op's / binary_boolean_kernel's texts.  The operand buffers are only
 * read during this call. */
typedef struct ah_filter_term {
  int32_t op; /* ah_cmp_op: AH_EQ .. AH_GT_EQ */
  const ah_array_view* lhs;
  int32_t lhs_is_scalar;
  const ah_array_view* rhs;
  int32_t rhs_is_scalar;
} ah_filter_term;
AH_API ah_status ah_filter_predicate_build_expr(ah_context* ctx, int32_t n_terms, const ah_filter_term* terms,
                                                const int32_t* joins /* n_terms - 1 ah_boolean_op values */,
                                                ah_filter_predicate** out);
/* `filter(values, <the expression>)` in one call: ah_filter_predicate_build_expr + ah_filter_predicate_apply. */
AH_API ah_status ah_filter_expr(ah_context* ctx, int32_t n_terms, const ah_filter_term* terms, const int32_t* joins,
                                const ah_array_view* values, ah_array_out* out);
/* Buffer::shrink_to_fit for fixed-width results whose buffers were allocated for the worst case (filters of small
 * batches): values / validity are copied into exact-size allocations; a result that fits or borrows is left alone. */
AH_API ah_status ah_array_shrink_to_fit(ah_context* ctx, ah_array_out* out);
AH_API int64_t ah_filter_predicate_count(const ah_filter_predicate* p); /* FilterPredicate::count :481 */
AH_API ah_status ah_filter_predicate_apply(ah_context* ctx, const ah_filter_predicate* p,
                                           const ah_array_view* values, ah_array_out* out);
AH_API void ah_filter_predicate_free(ah_context* ctx, ah_filter_predicate* p);
/* filter_record_batch (filter.rs:225): one predicate, n columns. */
AH_API ah_status ah_filter_record_batch(ah_context* ctx, int32_t n_columns,
                                        const ah_array_view* columns,
                                        const ah_array_view* predicate,
                                        ah_array_out* outs, int64_t* out_rows);

/* ------------------------------------------------------- coalesce support */
/* The two device primitives behind arrow_select::coalesce::BatchCoalescer
 * (arrow-select/src/coalesce.rs:148-330; host state machine in the mirrors):
 * InProgressPrimitiveArray::copy_rows_by_filter_from and ::copy_rows
 * (arrow-select/src/coalesce/primitive.rs).  dst_validity is the builder's bitmap: 8-byte
 * aligned, capacity >= target rows, zero in the range being appended to. */
AH_API ah_status ah_filter_predicate_apply_into(ah_context* ctx, const ah_filter_predicate* p,
                                                const ah_array_view* values, void* dst_values,
                                                uint8_t* dst_validity, int64_t dst_row_offset,
                                                int64_t* appended_nulls);
AH_API ah_status ah_copy_rows_into(ah_context* ctx, const ah_array_view* src, int64_t offset, int64_t len,
                                   void* dst_values, uint8_t* dst_validity, int64_t dst_row_offset,
                                   int64_t* appended_nulls);
/* The no-wait forms of the two calls above, for BatchCoalescer's per-batch loop: nothing is read back; the number of
 * NULL rows appended is added to the device word *nulls_acc (8-byte aligned device memory owned by the caller, one per
 * in-progress column).  ah_read_words fetches up to 200 such words in ONE host wait (optionally zeroing them) when the
 * in-progress batch is finished — NullBufferBuilder only needs the count then (coalesce/primitive.rs:94-106). */
AH_API ah_status ah_filter_predicate_apply_into_acc(ah_context* ctx, const ah_filter_predicate* p,
                                                    const ah_array_view* values, void* dst_values, uint8_t* dst_validity,
                                                    int64_t dst_row_offset, uint64_t* nulls_acc);
AH_API ah_status ah_copy_rows_into_acc(ah_context* ctx, const ah_array_view* src, int64_t offset, int64_t len,
                                       void* dst_values, uint8_t* dst_validity, int64_t dst_row_offset, uint64_t* nulls_acc);
AH_API ah_status ah_read_words(ah_context* ctx, uint64_t* dev_words, int32_t n, uint64_t* host_out, int32_t reset);

/* BatchCoalescer (arrow-select/src/coalesce.rs:148-700) as a native object: the reference's state machine — exact-size
 * output batches in input order, `biggest_coalesce_batch_size` bypass cases 1-3 (:296-420), push_batch_with_filter (:229),
 * finish_buffered_batch (:536), next_completed_batch (:566).  Fixed-width columns are InProgressPrimitiveArray
 * (coalesce/primitive.rs): filtered pushes scatter straight into the in-progress batch; no push waits for the GPU
 * except for the predicate's count; one wait per finished batch.  AH_BOOL, AH_UTF8 and AH_LARGE_UTF8 columns are
 * GenericInProgressArray (coalesce/generic.rs): filtered pieces are kept and concatenated when the batch is finished.  `tag` (any non-zero value naming the caller's batch)
 * comes back from ah_coalescer_next_completed_batch when that very batch was passed through untouched (large-batch
 * bypass; the push sets *bypassed = 1 so the caller knows to keep that batch alive): its outs are AH_OUT_BORROWED views of the
 * caller's buffers.  View column types: AH_NOT_YET_IMPLEMENTED.
 * Input lifetime: a push returns while its copies / scatters may still be running on the context's stream.  Buffers
 * that came from this context's allocator — the built-in pool, or the ah_context_set_allocator hook — may be released
 * at once (pool releases are stream-ordered; before a hook free the library drains the stream while no-wait work is in
 * flight); buffers the host allocated itself must stay alive until the stream has passed them (the fetch of a batch
 * finished after the push — ah_coalescer_next_completed_batch waits for that batch's work — or ah_synchronize).
 * Finishing a batch does not wait either: its null counts travel to pinned host words behind the last scatter and are
 * looked at when the batch is fetched. */
typedef struct ah_coalescer ah_coalescer;
AH_API ah_status ah_coalescer_create(ah_context* ctx, int32_t n_columns, const ah_type* types, int64_t target_batch_size,
                                     ah_coalescer** out);
AH_API void ah_coalescer_destroy(ah_context* ctx, ah_coalescer* co);
AH_API void ah_coalescer_set_biggest_coalesce_batch_size(ah_coalescer* co, int64_t limit /* < 0: none */);
AH_API int64_t ah_coalescer_buffered_rows(const ah_coalescer* co);
AH_API int32_t ah_coalescer_completed_count(const ah_coalescer* co);
AH_API ah_status ah_coalescer_push_batch(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows,
                                         uint64_t tag, int32_t* bypassed /* nullable */);
AH_API ah_status ah_coalescer_push_batch_with_filter(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns,
                                                     int64_t num_rows, const ah_array_view* filter, uint64_t tag,
                                                     int32_t* bypassed /* nullable */);
/* n filtered pushes in ONE call (columns: n x n_columns views, batch-major): identical output to n calls of
 * ah_coalescer_push_batch_with_filter, but the n predicate counts are read with a single host wait, so the GPU never
 * idles for a count round trip between batches — for hosts that have several batches queued. */
AH_API ah_status ah_coalescer_push_batches_with_filters(ah_context* ctx, ah_coalescer* co, int32_t n,
                                                        const ah_array_view* columns, const int64_t* num_rows,
                                                        const ah_array_view* filters, const uint64_t* tags /* nullable */,
                                                        int32_t* bypassed /* n entries, nullable */);
/* The same push in two halves, for a host that already holds the NEXT group of batches while this one is appended:
 * _begin checks the arguments, ENQUEUES the count passes of the n (<= 64) predicates and returns at once (their counts
 * travel to pinned words the coalescer owns; two groups may be in flight); _end waits for them and appends the batches
 * exactly as the one-call form does, then frees the handle.  Calling _begin of group g + 1 before _end of group g keeps
 * the GPU busy with group g's scatters while group g + 1's counts make their round trip to the host.  The view structs
 * are copied by _begin; the device buffers behind them must stay alive until _end.  Handles of one coalescer are ended
 * in the order they were begun, every handle is ended (also after an error elsewhere: _end is what frees it) and none
 * may be pending when the coalescer is destroyed.  At most two may be in flight with their counts travelling; a third
 * _begin simply builds its predicates synchronously. */
typedef struct ah_coalescer_push ah_coalescer_push;
AH_API ah_status ah_coalescer_push_batches_with_filters_begin(ah_context* ctx, ah_coalescer* co, int32_t n,
                                                              const ah_array_view* columns, const int64_t* num_rows,
                                                              const ah_array_view* filters, const uint64_t* tags /* nullable */,
                                                              ah_coalescer_push** handle);
AH_API ah_status ah_coalescer_push_batches_with_filters_end(ah_context* ctx, ah_coalescer* co, ah_coalescer_push* handle,
                                                            int32_t* bypassed /* n entries, nullable */);
/* Utf8View / BinaryView columns (InProgressByteViewArray, coalesce/byte_view.rs:39).  A view column is pushed as its
 * 16-byte views; the variadic data buffers stay with the host, as for filter / take.  Two calls carry what the
 * reference's builder does with them: BEFORE pushing, declare how many data buffers each view column of the next
 * `n_batches` batches has (n_batches x n_columns counts, batch-major; entries of non-view columns are ignored) — the
 * library shifts the buffer indices of the views it appends by the buffers the in-progress batch already references;
 * BEFORE fetching a completed batch, ask which inputs contributed rows to it — push sequence numbers, in order (every
 * pushed batch of the coalescer's life counts: 0, 1, 2, ...; a grouped push counts one per batch) — and attach the
 * concatenation of those inputs' buffer lists to the fetched view columns. */
AH_API ah_status ah_coalescer_declare_view_buffers(ah_context* ctx, ah_coalescer* co, int32_t n_batches, const int32_t* counts);
AH_API ah_status ah_coalescer_completed_batch_sources(ah_context* ctx, ah_coalescer* co, uint64_t* seqs, int32_t cap, int32_t* n);
/* push_batch_with_indices (coalesce.rs:289): take_record_batch(batch, indices), then push_batch of the result; indices
 * as for ah_take (unchecked: an out-of-range index is the reference's panic, AH_PANIC). */
AH_API ah_status ah_coalescer_push_batch_with_indices(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns,
                                                      int64_t num_rows, const ah_array_view* indices);
AH_API ah_status ah_coalescer_finish_buffered_batch(ah_context* ctx, ah_coalescer* co);
/* *num_rows = -1 when no batch is ready; outs[n_columns] are released with ah_array_release */
AH_API ah_status ah_coalescer_next_completed_batch(ah_context* ctx, ah_coalescer* co, ah_array_out* outs, int64_t* num_rows,
                                                   uint64_t* tag);
/* next_completed_batch for up to max_batches batches in one call: outs receives *n x n_columns results (batch-major),
 * num_rows and (optional) tags one entry per batch.  A grouped push of 8192-row batches (the reference's operating
 * point, coalesce.rs:172-173) completes thousands of output batches at once. */
AH_API ah_status ah_coalescer_next_completed_batches(ah_context* ctx, ah_coalescer* co, int32_t max_batches,
                                                     ah_array_out* outs, int64_t* num_rows, uint64_t* tags, int32_t* n);
/* ah_array_release for n results in one call */
AH_API void ah_arrays_release(ah_context* ctx, ah_array_out* outs, int64_t n);
/* Gives up a push begun with ah_coalescer_push_batches_with_filters_begin (waits for its count kernels, frees its
 * resources).  Its batches are not appended; their sequence numbers are consumed like those of a failed push. */
AH_API void ah_coalescer_push_abort(ah_context* ctx, ah_coalescer* co, ah_coalescer_push* handle);

/* ------------------------------------------------------------------ take */
/* arrow_select::take::take (arrow-select/src/take.rs:89).  indices.type is any
 * of the 8 integer types; i32/i64 are reinterpreted as u32/u64, 8/16-bit are
 * widened with `as u32` (take.rs:1030-1084).  check_bounds mirrors
 * TakeOptions (take.rs:388-394): AH_COMPUTE_ERROR with the text of take.rs:186;
 * unchecked OOB => AH_PANIC with the reference's panic text. */
AH_API ah_status ah_take(ah_context* ctx, const ah_array_view* values,
                         const ah_array_view* indices, int32_t check_bounds,
                         ah_array_out* out);

/* ----------------------------------------------------------------- arith */
typedef int32_t ah_arith_op;
enum {
  AH_ADD = 0, AH_ADD_WRAPPING = 1, AH_SUB = 2, AH_SUB_WRAPPING = 3,
  AH_MUL = 4, AH_MUL_WRAPPING = 5, AH_DIV = 6, AH_REM = 7,
  /* arrow_arith::bitwise::{bitwise_and, bitwise_or, bitwise_xor, bitwise_shift_left, bitwise_shift_right,
   * bitwise_and_not} and their `_scalar` forms (arrow-arith/src/bitwise.rs:42-205); integer types only; shift
   * counts are taken modulo the bit width (`wrapping_shl` / `wrapping_shr`) */
  AH_BIT_AND = 8, AH_BIT_OR = 9, AH_BIT_XOR = 10, AH_BIT_SHIFT_LEFT = 11, AH_BIT_SHIFT_RIGHT = 12, AH_BIT_AND_NOT = 13
};
/* arrow_arith::numeric::{add,add_wrapping,sub,sub_wrapping,mul,mul_wrapping,div,rem}
 * (arrow-arith/src/numeric.rs:36-81).  A Datum is (array, is_scalar)
 * (arrow-array/src/scalar.rs:78-98): scalars are length-1 views. */
AH_API ah_status ah_arith_binary(ah_context* ctx, ah_arith_op op,
                                 const ah_array_view* lhs, int32_t lhs_is_scalar,
                                 const ah_array_view* rhs, int32_t rhs_is_scalar,
                                 ah_array_out* out);
/* The same entry points when an operand is a temporal logical type (`arithmetic_op` numeric.rs:225-275, `timestamp_op`
 * :426-537, `duration_op` :877-895, `date_op` :898-932).  The result TYPE depends on the pair and is written to
 * `out_type`: Timestamp(u) - Timestamp(u) = Duration(u); Timestamp(u, tz) +- Duration(u) = Timestamp(u, tz);
 * Duration(u) +- Duration(u); Date64 - Date64 = Duration(ms); Date32 - Date32 = Duration(s) (`(l - r) * 86400`,
 * infallible); Duration + Timestamp swaps.  All of them are CHECKED i64 arithmetic, also through the *_WRAPPING ops
 * (the reference calls add_checked / sub_checked whatever the Op): AH_ARITHMETIC_OVERFLOW "Overflow happened on:
 * {l} + {r}".  Everything else is the reference's AH_INVALID_ARGUMENT text ("Invalid timestamp arithmetic operation:
 * Timestamp(s) * Duration(s)", "Invalid arithmetic operation: Int64 + Timestamp(ms)", ...); the Interval arms are
 * AH_NOT_YET_IMPLEMENTED.  Plain numeric pairs are forwarded to ah_arith_binary.
 * Decimal128 (both operands AH_DT_DECIMAL128 over AH_FIXED16 values; `decimal_op` numeric.rs:971-1103): the scales are
 * aligned with checked powers of ten and every row is `l.mul_checked(l_mul)? op (r.mul_checked(r_mul)?)` over i128;
 * out_type carries the Hive-rule result type (add / sub: scale max(s1, s2), precision max(s1, s2) + max(p1 - s1,
 * p2 - s2) + 1; mul: p1 + p2 + 1, s1 + s2 — "Output scale of .. would exceed max scale of 38" otherwise; div: scale
 * s1 + 4; rem: scale max(s1, s2); precision capped at 38).  Errors are the reference's: "Overflow happened on: {a} *
 * {b}" for the row that fails first, "Overflow happened on: 10 ^ 39", "Divide by zero error". */
AH_API ah_status ah_arith_with_types(ah_context* ctx, ah_arith_op op, const ah_array_view* lhs, int32_t lhs_is_scalar,
                                     const ah_data_type* lhs_type, const ah_array_view* rhs, int32_t rhs_is_scalar,
                                     const ah_data_type* rhs_type, ah_array_out* out, ah_data_type* out_type);
/* bitwise_not (arrow-arith/src/bitwise.rs:113) */
AH_API ah_status ah_bitwise_not(ah_context* ctx, const ah_array_view* values, ah_array_out* out);
/* neg / neg_wrapping (numeric.rs:103,181) */
AH_API ah_status ah_arith_neg(ah_context* ctx, const ah_array_view* values, int32_t wrapping,
                              ah_array_out* out);

/* ------------------------------------------------------------------- cmp */
typedef int32_t ah_cmp_op;
enum {
  AH_EQ = 0, AH_NEQ = 1, AH_LT = 2, AH_LT_EQ = 3, AH_GT = 4, AH_GT_EQ = 5,
  AH_DISTINCT = 6, AH_NOT_DISTINCT = 7
};
/* arrow_ord::cmp::{eq,neq,lt,lt_eq,gt,gt_eq,distinct,not_distinct}
 * (arrow-ord/src/cmp.rs:79-202).  Result type AH_BOOL; floats compare in IEEE
 * totalOrder (arrow-array/src/arithmetic.rs:400-410).  AH_FIXED16 operands compare as i128 (Decimal128: the host
 * checks that precision and scale agree, as compare_op does); IntervalMonthDayNano shares the layout but not the
 * order and must not be passed. */
AH_API ah_status ah_compare(ah_context* ctx, ah_cmp_op op,
                            const ah_array_view* lhs, int32_t lhs_is_scalar,
                            const ah_array_view* rhs, int32_t rhs_is_scalar,
                            ah_array_out* out);
/* ah_compare with compare_op's LOGICAL-type rule (cmp.rs:243-264): the two sides' DataTypes must be equal (Decimal128
 * precision and scale, time units, zones included), else AH_INVALID_ARGUMENT "Invalid comparison operation: {l} {op} {r}";
 * the values then compare as their physical type. */
AH_API ah_status ah_compare_with_types(ah_context* ctx, ah_cmp_op op, const ah_array_view* lhs, int32_t lhs_is_scalar,
                                       const ah_data_type* lhs_type, const ah_array_view* rhs, int32_t rhs_is_scalar,
                                       const ah_data_type* rhs_type, ah_array_out* out);

/* --------------------------------------------------------------- boolean */
typedef int32_t ah_boolean_op;
enum {
  AH_BOOL_AND = 0, AH_BOOL_OR = 1, AH_BOOL_AND_NOT = 2, AH_BOOL_AND_KLEENE = 3, AH_BOOL_OR_KLEENE = 4,
  AH_BOOL_NOT = 10, AH_BOOL_IS_NULL = 11, AH_BOOL_IS_NOT_NULL = 12
};
/* arrow_arith::boolean::{and,or,and_not,and_kleene,or_kleene} (arrow-arith/src/boolean.rs:60-300):
 * both inputs AH_BOOL of equal length, else AH_COMPUTE_ERROR "Cannot perform bitwise operation on
 * arrays of different length". */
AH_API ah_status ah_boolean_binary(ah_context* ctx, ah_boolean_op op, const ah_array_view* left,
                                   const ah_array_view* right, ah_array_out* out);
/* not (boolean.rs:310, AH_BOOL input), is_null / is_not_null (:327,:347, any type; the result
 * never carries a null buffer). */
AH_API ah_status ah_boolean_unary(ah_context* ctx, ah_boolean_op op, const ah_array_view* values,
                                  ah_array_out* out);

/* arrow_select::nullif::nullif (arrow-select/src/nullif.rs:60): values shared with `left`
 * (AH_OUT_BORROWED_VALUES), new validity = left_validity & !(right_values & right_validity). */
AH_API ah_status ah_nullif(ah_context* ctx, const ah_array_view* left, const ah_array_view* right,
                           ah_array_out* out);

/* ------------------------------------------------- string predicates / length */
/* arrow_string::like::{like, nlike, starts_with, ends_with, contains} (arrow-string/src/like.rs:83-205) on
 * AH_UTF8 / AH_LARGE_UTF8 values against a SCALAR pattern of the same type (a length-1 array, Datum::get()):
 * `%` = any run of characters, `_` = exactly one character, `\x` = literal x, a trailing `\` = a literal
 * backslash (Predicate::like / regex_like, arrow-string/src/predicate.rs:44-306); starts_with / ends_with /
 * contains take the needle literally.  Boolean result, input nulls cloned (BooleanArray::from_unary), a null
 * pattern gives an all-null result (like.rs:314).  Per-row pattern arrays and the case-insensitive forms
 * (ilike: Unicode case folding) are AH_NOT_YET_IMPLEMENTED. */
typedef int32_t ah_like_op;
enum { AH_LIKE = 0, AH_NLIKE = 1, AH_STARTS_WITH = 2, AH_ENDS_WITH = 3, AH_CONTAINS = 4 };
AH_API ah_status ah_string_like(ah_context* ctx, ah_like_op op, const ah_array_view* values,
                                const ah_array_view* pattern, int32_t pattern_is_scalar, ah_array_out* out);
/* arrow_string::length::{length, bit_length} (arrow-string/src/length.rs:58,:130) for AH_UTF8 (Int32 result) /
 * AH_LARGE_UTF8 (Int64): byte length of every value (x8 when `bits`), nulls cloned. */
AH_API ah_status ah_string_length(ah_context* ctx, const ah_array_view* values, int32_t bits, ah_array_out* out);

/* ------------------------------------------------------------------ cast */
/* arrow_cast::cast_with_options (arrow-cast/src/cast/mod.rs:790), restricted to
 * numeric<->numeric (mod.rs:1578-1697 via cast_numeric_arrays :2550) and
 * Float64/Float32/integers -> Utf8 / LargeUtf8 (mod.rs:1552-1553 via
 * value_to_string, cast/string.rs:21-39), Boolean <-> numeric (mod.rs:1243-1290:
 * `value != 0`; true -> 1, false -> 0), and Utf8 / LargeUtf8 -> integers / Float32 / Float64
 * (parse_string, cast/string.rs:66-120 -> Parser::parse, parse.rs:446-528: text that does not parse becomes
 * null in safe mode — a null buffer is always attached — and AH_CAST_ERROR "Cannot cast string '..' to value
 * of <T> type" for the first such row otherwise; floats are correctly rounded).  safe mirrors CastOptions.safe. */
AH_API ah_status ah_cast(ah_context* ctx, const ah_array_view* values, ah_type to_type,
                         int32_t safe, ah_array_out* out);
AH_API int32_t ah_can_cast_types(ah_type from, ah_type to); /* cast/mod.rs:115 subset */

/* arrow_cast::cast_with_options for (from, to) pairs where either side is AH_DT_*; everything else is forwarded to
 * ah_cast.  `values->type` must be the physical layout of `from` (AH_INT32 for Date32 / Time32, AH_INT64 for the
 * rest).  Arm by arm as the reference: reinterpreting arms clone; unit up-scaling is `checked_mul` (safe: overflow
 * becomes null and a null buffer is always attached, unary_opt; unsafe: AH_ARITHMETIC_OVERFLOW "Overflow happened
 * on: {v} * {k}", or the wrapping `x * k` where the reference writes it unchecked); down-scaling is truncating `/`;
 * Date64 -> Date32 is `i32::try_from(x / 86_400_000)` (AH_CAST_ERROR "Cannot cast Date64 value {x} to Date32 without
 * overflow"); Timestamp -> Date32 / Time32 / Time64 go through chrono's calendar (0.4.45, not vendored: floor
 * division into days since 1970-01-01, valid for NaiveDate::MIN..=MAX = -262143-01-01..=+262142-12-31, local to
 * the source zone) and fail in BOTH modes with "Cannot convert {type} {x} to datetime" / "Failed to create naive
 * time with {type} {x}" (try_unary, :633-659, :615-631); a zone-less Timestamp cast to a zoned one keeps the wall
 * clock (adjust_timestamp_to_timezone :2629; unsafe failure: "Cannot cast timezone to different timezone").
 * Two AH_DT_DECIMAL128 descriptors (values AH_FIXED16) select cast_decimal_to_decimal_same_type (cast/decimal.rs:448-489):
 * rescale by 10^(scale difference), rounding half away from zero when the scale shrinks; a value that does not fit
 * the output precision is null in safe mode and, otherwise, AH_INVALID_ARGUMENT "{v} is too large to store in a
 * Decimal128 of precision {p}. Max is {max}" / AH_CAST_ERROR "Cannot cast to Decimal128(p, s). Overflowing on {x}".
 * A plain integer `from` with an AH_DT_DECIMAL128 `to` is cast_integer_to_decimal (cast/mod.rs:366-443): v * 10^scale
 * checked (AH_ARITHMETIC_OVERFLOW "Overflow happened on: {v} * {10^s}"), or v / 10^|scale| in the source type for a
 * negative scale, then the precision test.  The other decimal casts are AH_NOT_YET_IMPLEMENTED. */
AH_API ah_status ah_cast_with_types(ah_context* ctx, const ah_array_view* values, const ah_data_type* from,
                                    const ah_data_type* to, int32_t safe, ah_array_out* out);
/* cast(cast(values, types[0]), types[1]) ... as ONE call (BASELINE configs[3]: Int64 -> Float64 -> Utf8).  Results are
 * byte-identical to the step-by-step casts; what is materialised in between is the library's business:
 * Int64 -> Float64 -> Utf8 / LargeUtf8 formats straight from the Int64 column, never building the Float64 array. */
AH_API ah_status ah_cast_chain(ah_context* ctx, const ah_array_view* values, int32_t n_types, const ah_type* types,
                               int32_t safe, ah_array_out* out);
AH_API int32_t ah_can_cast_data_types(const ah_data_type* from, const ah_data_type* to); /* cast/mod.rs:115 */

/* ---------------------------------------------------------------- concat */
/* arrow_select::concat::concat for primitives / booleans (concat.rs:334-343,
 * :495) — also the multi-GPU reassembly primitive (bit-shifted bitmap merge,
 * reference analogue arrow-buffer/src/util/bit_mask.rs:33 set_bits). */
AH_API ah_status ah_concat(ah_context* ctx, int32_t n, const ah_array_view* pieces,
                           ah_array_out* out);
/* arrow_select::window::shift (arrow-select/src/window.rs:56-80): positive offsets shift right, negative left,
 * vacated slots are null; offset 0 returns the input's buffers (AH_OUT_BORROWED); |offset| >= length gives an
 * all-null array of the same type. */
AH_API ah_status ah_shift(ah_context* ctx, const ah_array_view* values, int64_t offset, ah_array_out* out);
/* OR `len` bits from (src, src_bit_offset) into dst at dst_bit_offset; dst bits
 * in range must be zero.  Returns the number of set bits copied. */
AH_API ah_status ah_bitmap_set_bits(ah_context* ctx, uint8_t* dst, int64_t dst_bit_offset,
                                    const uint8_t* src, int64_t src_bit_offset, int64_t len,
                                    int64_t* set_bits);

/* ---------------------------------------------------- multi-GPU exchange (RCCL over xGMI) */
/* The reference has no parallelism; the north star shards a RecordBatch by contiguous row range over the GPUs of a
 * node (RecordBatch::slice semantics, arrow-array/src/record_batch.rs:681), one process per GPU, and reassembles the
 * per-shard kernel results as their in-order concatenation — arrow_select::concat (arrow-select/src/concat.rs:334-343,
 * :495) across ranks, validity merged like bit_mask::set_bits (arrow-buffer/src/util/bit_mask.rs:33).
 *
 *   rank 0:      ah_comm_unique_id(ctx, id)            -> ship the 128 bytes to every rank by any means
 *   every rank:  ah_comm_create(ctx, rank, world, id)   (ncclCommInitRank; RCCL is dlopen'ed on first use)
 *                ah_all_gatherv / ah_all_gather_columns on results of ah_filter & co.
 *
 * One ncclAllGather of the per-rank counts (the call's one host wait before the exchange), ONE grouped
 * ncclSend/ncclRecv that lands every peer's values (and string bytes) at their final offset, ONE kernel that merges
 * every staged bitmap (validity of every column, Boolean value bits: concat_boolean, concat.rs:345) and one
 * offset-rebase kernel per Utf8 / LargeUtf8 column (concat_bytes, concat.rs:355; a Utf8 total past i32::MAX bytes is
 * AH_OFFSET_OVERFLOW_ERROR on every rank, before anything moves).  Column types: fixed-width, AH_BOOL, AH_UTF8,
 * AH_LARGE_UTF8 (views are refused like ah_concat refuses them).  RCCL's asynchronous error state is checked after
 * the collective (AH_COMM_ERROR).
 * Failing together: the count payload carries a per-rank status word and every column's type, so a rank whose
 * arguments are bad still takes part in the count exchange and EVERY rank returns an error (its own, or
 * AH_COMM_ERROR "rank r failed before the exchange" / AH_INVALID_ARGUMENT for a schema mismatch) instead of
 * leaving peers blocked.  A rank that fails between the count exchange and the grouped exchange (out of memory)
 * aborts the communicator (ncclCommAbort): its peers see an asynchronous RCCL error, and every later call on that
 * communicator is AH_COMM_ERROR.
 * `id == NULL` with world == 1 makes a communicator that never touches RCCL (copies only). */
#define AH_COMM_ID_BYTES 128
typedef struct ah_comm ah_comm;
typedef struct ah_exchange_stats {
  int32_t peers;              /* world - 1 */
  int64_t bytes_to_each_peer; /* values + packed validity this rank pushed to EACH peer (one xGMI link each) */
  int64_t bytes_received;     /* from all peers together */
  double counts_ms;           /* host time until the counts were known */
  double total_ms;            /* host time of the whole call */
} ah_exchange_stats;
AH_API ah_status ah_comm_unique_id(ah_context* ctx, uint8_t* id /* AH_COMM_ID_BYTES */);
AH_API ah_status ah_comm_create(ah_context* ctx, int32_t rank, int32_t world, const uint8_t* id, ah_comm** out);
AH_API void ah_comm_destroy(ah_context* ctx, ah_comm* comm);
AH_API int32_t ah_comm_rank(const ah_comm* comm);
AH_API int32_t ah_comm_world(const ah_comm* comm);
/* stream-ordered barrier over all ranks / element-wise max of up to 32 doubles (bench timing: MAX over ranks) */
AH_API ah_status ah_comm_barrier(ah_context* ctx, ah_comm* comm);
AH_API ah_status ah_comm_allreduce_max_f64(ah_context* ctx, ah_comm* comm, double* values, int32_t n);
/* out = concat(rank 0's array, rank 1's array, ...) on every rank; `stats` may be NULL */
AH_API ah_status ah_all_gatherv(ah_context* ctx, ah_comm* comm, const ah_array_view* local, ah_array_out* out,
                                ah_exchange_stats* stats);
/* the same for the n_columns (<= 16) columns of a RecordBatch shard: one count exchange and one grouped exchange
 * for all of them (== concat_batches of the shard results, concat.rs:607) */
AH_API ah_status ah_all_gather_columns(ah_context* ctx, ah_comm* comm, int32_t n_columns, const ah_array_view* columns,
                                       ah_array_out* outs, ah_exchange_stats* stats);
/* The two halves of ah_all_gather_columns for a host that has other work to overlap with the exchange (a `take` on a
 * second context = a second stream, bench.py's filter_take at N > 1).  _begin does the count exchange (one host wait:
 * output sizes depend on it), allocates the outputs, enqueues the grouped exchange and the merge kernels on the
 * context's stream and returns; _end waits for the stream, checks RCCL's asynchronous error state and hands over
 * the outputs (n_columns of them) and the statistics.  The `columns` buffers must stay alive until _end; every _begin
 * must be followed by exactly one _end on the same context and communicator (on failure _begin leaves *handle NULL
 * and nothing to end); exchanges of one communicator do not nest. */
typedef struct ah_exchange ah_exchange;
AH_API ah_status ah_all_gather_columns_begin(ah_context* ctx, ah_comm* comm, int32_t n_columns, const ah_array_view* columns,
                                             ah_exchange** handle);
AH_API ah_status ah_all_gather_columns_end(ah_context* ctx, ah_comm* comm, ah_exchange* handle, ah_array_out* outs,
                                           ah_exchange_stats* stats);
/* The merge primitive on its own: dst (8-byte aligned, ceil(total / 64) words, written in full) = the concatenation of
 * `n` bit-packed pieces (pieces[i], bit_offsets[i], lens[i]); pieces[i] == NULL reads as all ones; bit_offsets may
 * be NULL (all zero).  bitmap concat of arrow-select/src/concat.rs:300-330 / bit_mask.rs:33 in ONE launch. */
AH_API ah_status ah_bitmap_concat(ah_context* ctx, int32_t n, const uint8_t* const* pieces, const int64_t* bit_offsets,
                                  const int64_t* lens, uint8_t* dst, int64_t* total_rows);

/* ------------------------------------------------------------- aggregate */
/* arrow_arith::aggregate (arrow-arith/src/aggregate.rs): `sum` :943, `sum_checked` :897,
 * `product` :953, `product_checked` :963, `min` :1012, `max` :1027, `bit_and/bit_or/bit_xor`
 * :776-873; on AH_BOOL, AH_AGG_MIN / AH_AGG_MAX are `min_boolean` / `max_boolean` (:372,:430)
 * = `bool_and` / `bool_or` (:880-889).  `Option<T::Native>`: is_valid == 0 is `None`
 * (empty or all-null input).  Integer sum/product wrap; min/max use the total order for floats
 * (NaN above +inf, -NaN below -inf); the *_checked forms fail like the reference's sequential
 * loop, i.e. when a PREFIX overflows, with its "Overflow happened on: {acc} + {value}" text.
 * Float sums/products are a fixed tree reduction (the reference's own association order depends
 * on its compile-time vector width, :300-307). */
typedef int32_t ah_agg_op;
enum {
  AH_AGG_SUM = 0, AH_AGG_SUM_CHECKED = 1, AH_AGG_PRODUCT = 2, AH_AGG_PRODUCT_CHECKED = 3,
  AH_AGG_MIN = 4, AH_AGG_MAX = 5, AH_AGG_BIT_AND = 6, AH_AGG_BIT_OR = 7, AH_AGG_BIT_XOR = 8
};
typedef struct ah_scalar {
  ah_type type;
  int32_t is_valid;  /* 0 = None */
  uint8_t bytes[32]; /* little-endian native value (AH_BOOL: one byte 0/1) */
} ah_scalar;
AH_API ah_status ah_aggregate(ah_context* ctx, ah_agg_op op, const ah_array_view* values, ah_scalar* out);

/* ------------------------------------------------------------- utilities */
/* BooleanBuffer::count_set_bits (arrow-buffer/src/buffer/boolean.rs) on device. */
AH_API ah_status ah_count_set_bits(ah_context* ctx, const uint8_t* bits, int64_t bit_offset,
                                   int64_t len, int64_t* count);
/* Kernel timing with HIP events on the context stream (for bench.py's roofline
 * leg): when enabled, every launch of the named hot kernels is bracketed by
 * hipEventRecord on the launch stream. */
AH_API void ah_profile_enable(ah_context* ctx, int32_t on);
AH_API void ah_profile_reset(ah_context* ctx);
AH_API ah_status ah_profile_get(ah_context* ctx, const char* kernel, double* total_ms,
                                int64_t* launches);

/* Counter-based synthetic data: splitmix64 of (seed, row) so
 * the CPU oracle reproduces any window bit-for-bit.  Test/bench support only. */
AH_API ah_status ah_gen_uniform_i64(ah_context* ctx, int64_t* dst, int64_t n, uint64_t seed,
                                    int64_t lo, int64_t hi_inclusive, int64_t row0);
AH_API ah_status ah_gen_uniform_i32(ah_context* ctx, int32_t* dst, int64_t n, uint64_t seed,
                                    int64_t row0);
AH_API ah_status ah_gen_uniform_f64(ah_context* ctx, double* dst, int64_t n, uint64_t seed,
                                    double lo, double hi, int64_t row0);
AH_API ah_status ah_gen_uniform_u32(ah_context* ctx, uint32_t* dst, int64_t n, uint64_t seed,
                                    uint32_t bound, int64_t row0);
AH_API ah_status ah_gen_uniform_f32(ah_context* ctx, float* dst, int64_t n, uint64_t seed,
                                    float lo, float hi, int64_t row0);
/* byte_width 1 or 2: the low bytes of splitmix64(seed, row) (full-range Int8 / Int16 patterns) */
AH_API ah_status ah_gen_uniform_small(ah_context* ctx, void* dst, int32_t byte_width, int64_t n,
                                      uint64_t seed, int64_t row0);
/* dst[i] = start + i (row positions, e.g. to turn a predicate into take indices) */
AH_API ah_status ah_gen_iota_u32(ah_context* ctx, uint32_t* dst, int64_t n, uint32_t start);
/* Bernoulli(p_true) bits, LSB-first, written at bit offset 0 of dst
 * (ceil(n/64)*8 bytes; padding bits zero). */
AH_API ah_status ah_gen_bernoulli_bits(ah_context* ctx, uint8_t* dst, int64_t n, uint64_t seed,
                                       double p_true, int64_t row0);
/* dst[i] = 0 where validity bit i is 0 (null slots hold 0 like the ref's
 * FromIterator<Option<T>> bench arrays, arrow/src/util/bench_util.rs:45-60) */
AH_API ah_status ah_zero_null_slots(ah_context* ctx, void* values, int32_t byte_width,
                                    const uint8_t* validity, int64_t n);

/* ------------------------------------------------ Arrow C Data Interface */
/* The second ABI-stable surface of the reference (arrow-data/src/ffi.rs:37-66
 * `FFI_ArrowArray`, arrow-schema/src/ffi.rs:76-98 `FFI_ArrowSchema`).  The struct
 * definitions are the Arrow specification's, under its own include guard so
 * they coexist with any other copy in the host program. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

/* `DataType::try_from(&FFI_ArrowSchema)` (arrow-schema/src/ffi.rs:492-700) reduced
 * to the physical layout: "l" / "tsu:UTC" / "tDn" -> AH_INT64, "d:38,10" ->
 * AH_FIXED16, "u" -> AH_UTF8 ...  The logical type stays with the host, which hands the
 * same format string back to ah_export_c_data.  Unknown formats: AH_C_DATA_INTERFACE
 * with the reference text; layouts this library has no kernels for (nested,
 * dictionary, views, binary): AH_NOT_YET_IMPLEMENTED. */
AH_API ah_status ah_type_from_format(ah_context* ctx, const char* format, ah_type* out);
/* Default format string of a physical type ("l" for AH_INT64, "d:38,0" for AH_FIXED16...). */
AH_API const char* ah_format_of_type(ah_type t);

/* `from_ffi(array, &schema)` (arrow-array/src/ffi.rs:237-254): a HOST-resident
 * C-Data array becomes a device-resident array (one H2D copy per buffer on the
 * context stream; the rows [offset, offset+length) only).  `array` and `schema`
 * stay owned by the caller (borrowed like every other input); the result is
 * owned like any kernel output.  null_count -1 is counted on the device
 * (`null_count_opt`, arrow-data/src/ffi.rs:327); a present validity buffer is
 * kept even when null_count == 0, an absent one stays absent. */
AH_API ah_status ah_import_c_data(ah_context* ctx, const struct ArrowArray* array,
                                  const struct ArrowSchema* schema, ah_array_out* out);
/* `to_ffi(&data)` (arrow-array/src/ffi.rs:231-235): device array -> host-resident
 * C-Data structs whose release callbacks free the host copies.  `format` NULL =
 * the physical type's default; otherwise it must map to `values->type`.
 * Like `FFI_ArrowArray::new` (arrow-data/src/ffi.rs:133-212) the null buffer is
 * re-aligned to the exported offset (always 0 here), n_buffers follows the spec
 * (2, or 3 for strings) and buffers[0] is NULL when there is no null buffer. */
AH_API ah_status ah_export_c_data(ah_context* ctx, const ah_array_view* values, const char* format,
                                  struct ArrowArray* out_array, struct ArrowSchema* out_schema);

/* -------------------------------------------- Arrow C Device Data Interface */
/* The device-aware extension of the C Data Interface (Arrow specification, "C Device data interface"): the same
 * ArrowArray, whose buffer pointers are DEVICE pointers, plus the device it lives on and an optional event to wait
 * for.  arrow-rs has no counterpart (its ffi module is host-only, arrow-array/src/ffi.rs); this is the zero-copy
 * hand-off between this library and any other ROCm producer / consumer. */
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* ROCm: hipEvent_t*, or NULL when the data is already complete */
  int64_t reserved[3];
};
#endif
/* Zero-copy export of a device array.  Values / offsets / data buffers are exported by pointer; a validity (or
 * Boolean value) bitmap that does not start on bit 0 is re-aligned into a fresh device buffer first
 * (`align_nulls`, arrow-data/src/ffi.rs:104), so `array.offset` is always 0.  If `owned` is non-NULL its buffers
 * MOVE into the exported struct (the release callback frees them through the context, which must outlive it) and
 * `*owned` is cleared; otherwise the export borrows `values` and the caller keeps the source alive.
 * sync_event is NULL: every entry point of this library is synchronous at return. */
AH_API ah_status ah_export_c_device_data(ah_context* ctx, const ah_array_view* values, ah_array_out* owned,
                                         const char* format, struct ArrowDeviceArray* out_array,
                                         struct ArrowSchema* out_schema);
/* Zero-copy import: a borrowed view of a ROCm-resident ArrowDeviceArray on this context's device (the producer
 * keeps ownership; waits for `sync_event` on the context stream when one is given). */
AH_API ah_status ah_import_c_device_data(ah_context* ctx, const struct ArrowDeviceArray* array,
                                         const struct ArrowSchema* schema, ah_array_view* out_view);

/* ------------------------------------------------------------ interleave */
/* arrow_select::interleave::interleave (arrow-select/src/interleave.rs:74): out[i] = arrays[array_index[i]][row_index[i]]
 * — `take` across several arrays of one type (merge, repartition).  The reference's `&[(usize, usize)]` is two
 * UInt32 device arrays here.  Fixed-width and Boolean layouts; a null buffer iff some input has nulls; an
 * out-of-range pair is the reference's slice-indexing panic (AH_PANIC). */
AH_API ah_status ah_interleave(ah_context* ctx, int32_t n_arrays, const ah_array_view* arrays,
                               const ah_array_view* array_index, const ah_array_view* row_index, ah_array_out* out);

/* ------------------------------------------------------------------- zip */
/* arrow_select::zip::zip (arrow-select/src/zip.rs:99): out[i] = mask[i] ? truthy[i] : falsy[i]; a null mask row
 * selects `falsy`; either side may be a length-1 scalar (`Datum::get()`).  Fixed-width and Boolean layouts.
 * The result carries a null buffer iff an input has nulls.  Error texts of :115-140. */
AH_API ah_status ah_zip(ah_context* ctx, const ah_array_view* mask, const ah_array_view* truthy, int32_t truthy_is_scalar,
                        const ah_array_view* falsy, int32_t falsy_is_scalar, ah_array_out* out);

/* ------------------------------------------------------------------ sort */
/* arrow_ord::sort::sort_to_indices (arrow-ord/src/sort.rs:276): UInt32 row numbers that order `values`
 * (`SortOptions { descending, nulls_first }`, arrow-schema/src/lib.rs:87; `limit` < 0 = None) — the usual
 * producer of the indices `ah_take` consumes (`sort` / `sort_limit` = take(values, sort_to_indices(...)),
 * sort.rs:56-177).  Integers, floats (IEEE totalOrder, like `lt`), Boolean and Utf8 / LargeUtf8 (bytewise, a
 * proper prefix first: `sort_bytes`); null rows keep ascending row
 * order at the front or the back (`sort_impl` :639-672).  The reference uses an unstable sort, so the order of
 * equal keys is unspecified there: this implementation is STABLE (equal keys in ascending row order, also when
 * descending), which is the order every tie in the reference's own tests shows. */
AH_API ah_status ah_sort_to_indices(ah_context* ctx, const ah_array_view* values, int32_t descending,
                                    int32_t nulls_first, int64_t limit, ah_array_out* out);

/* arrow_ord::rank::rank (arrow-ord/src/rank.rs:58): UInt32 rank of every row in the sorted order (1-based), equal
 * values sharing the highest of their ranks, null rows sharing one rank before (nulls_first) or after all values.
 * Never null.  Integers, Float32 / Float64 (totalOrder equality), Boolean and Utf8 / LargeUtf8; the reference's
 * Vec<u32> result is
 * the values buffer of the output. */
AH_API ah_status ah_rank(ah_context* ctx, const ah_array_view* values, int32_t descending, int32_t nulls_first,
                         ah_array_out* out);

/* arrow_ord::sort::lexsort_to_indices (sort.rs:939): rows ordered by cols[0], ties by cols[1], ... — a chain of
 * stable single-column sorts from the last column to the first.  One column delegates to ah_sort_to_indices
 * (:949); rows equal on every column stay in ascending row order.  `lexsort` = ah_take per column. */
AH_API ah_status ah_lexsort_to_indices(ah_context* ctx, int32_t n_cols, const ah_array_view* cols,
                                       const int32_t* descending, const int32_t* nulls_first, int64_t limit,
                                       ah_array_out* out);

/* ---------------------------------------------------------- row selection */
/* parquet `RowSelection`, bitmap-backed form (parquet/src/arrow/arrow_reader/selection/): the structure the
 * parquet reader's row-filter loop (arrow_reader/read_plan.rs) builds from predicate results and chains with
 * `and_then` — the largest in-tree caller of `filter`.  Masks are AH_BOOL views WITHOUT nulls (a null count
 * > 0 is the reference's `assert_eq!(filter.null_count(), 0)` panic, mod.rs:318).
 *  and_then      : `RowSelection::and_then` on masks (algebra.rs:392-436): out[i] = mask[i] & other[rank(i)];
 *                  other->length must equal the number of set bits of mask (reference panic texts otherwise);
 *                  all-true `other` returns the input zero-copy (AH_OUT_BORROWED).
 *  combine       : op 0 `intersection`, 1 `union` (algebra.rs:267-349): over the common prefix, the longer
 *                  side's tail passes through.
 *  boundaries    : the RLE form (`mask_to_selectors`, boolean.rs:172-190) as ascending Int64 positions p with
 *                  bit[p] != bit[p-1] (bit[-1] = 0): runs alternate skip/select starting with a skip run that
 *                  is empty when the first boundary is 0.
 *  from_boundaries: the inverse (`From<Vec<RowSelector>>`, `from_consecutive_ranges` mod.rs:326).
 *  find_nth_set_bit: `BooleanBuffer::find_nth_set_bit_position(start, n)` (arrow-buffer/src/buffer/
 *                  boolean.rs:445): one past the n-th (1-based) set bit at or after start, else length — the
 *                  primitive under `offset` / `limit` / `trim` (boolean.rs:281-309). */
AH_API ah_status ah_selection_and_then(ah_context* ctx, const ah_array_view* mask, const ah_array_view* other,
                                       ah_array_out* out);
AH_API ah_status ah_selection_combine(ah_context* ctx, int32_t op, const ah_array_view* l, const ah_array_view* r,
                                      ah_array_out* out);
AH_API ah_status ah_selection_boundaries(ah_context* ctx, const ah_array_view* mask, ah_array_out* out);
AH_API ah_status ah_selection_from_boundaries(ah_context* ctx, const ah_array_view* bounds, int64_t total_rows,
                                              ah_array_out* out);
AH_API ah_status ah_selection_find_nth_set_bit(ah_context* ctx, const ah_array_view* mask, int64_t start, int64_t n,
                                               int64_t* pos);

/* ------------------------------------------------------------- Arrow IPC */
/* Record-batch framing of the Arrow IPC format (arrow-ipc/src/writer.rs, reader.rs; format/Message.fbs,
 * format/Schema.fbs) for batches whose buffers live in HBM.  A message is a small host-side metadata block
 * ([0xFFFFFFFF][padded length][flatbuffer][padding], writer.rs:138-222) plus a BODY; here the body is ONE
 * contiguous device buffer, so a batch moves as a single transfer (D2H, or one RCCL send between GPUs) and an
 * incoming body is decoded into zero-copy views.  The schema language is the C Data format string
 * (ah_type_from_format).  Scope: the fixed-width, Boolean and (Large)Utf8/Binary layouts this library has
 * kernels for; dictionaries, nested types, views and compressed bodies are AH_NOT_YET_IMPLEMENTED. */
typedef struct ah_ipc_field {
  const char* name;
  const char* format; /* C Data format string: "l", "g", "tsu:UTC", "d:38,10", "u", ... */
  int32_t nullable;
} ah_ipc_field;
/* Message header ordinals of format/Message.fbs */
enum { AH_IPC_SCHEMA = 1, AH_IPC_DICTIONARY_BATCH = 2, AH_IPC_RECORD_BATCH = 3 };

/* File.fbs `struct Block`: where a message sits in an IPC FILE (offset of its continuation marker from the start
 * of the file, framed metadata length, body length). */
typedef struct ah_ipc_block {
  int64_t offset;
  int32_t meta_data_length;
  int32_t reserved_; /* struct padding, zero on the wire */
  int64_t body_length;
} ah_ipc_block;

/* `IpcDataGenerator::schema_to_bytes` (writer.rs): framed Schema message; free with ah_host_free. */
AH_API ah_status ah_ipc_schema_message(ah_context* ctx, int32_t n_fields, const ah_ipc_field* fields,
                                       int32_t alignment, uint8_t** out, int64_t* out_len);
/* Framed Schema message -> fields.  `*fields` is one allocation (array + strings): ah_host_free(*fields). */
AH_API ah_status ah_ipc_decode_schema(ah_context* ctx, const uint8_t* msg, int64_t len, int32_t* n_fields,
                                      ah_ipc_field** fields);
/* `record_batch_to_bytes` (writer.rs:1006): framed RecordBatch metadata on the host (ah_host_free) and the body
 * in HBM (owned through the context allocator: ah_device_free / the host allocator hook).  Per column, as
 * `write_array_data` (:2364): validity always present (all ones when the array has no null buffer), bitmaps
 * re-aligned to bit 0, values cut to the slice, string offsets rebased to 0 and data cut to the referenced
 * range; every buffer padded to `alignment` (8/16/32/64; the reference defaults to 64). */
AH_API ah_status ah_ipc_encode_batch(ah_context* ctx, int32_t n_cols, const ah_array_view* cols, int64_t num_rows,
                                     int32_t alignment, uint8_t** out_meta, int64_t* out_meta_len,
                                     void** out_body, int64_t* out_body_len);
/* `RecordBatchDecoder` (reader.rs:88-300): framed metadata (host) + body (DEVICE pointer) -> one array per
 * field.  Results are AH_OUT_BORROWED views into `body` (keep it alive) unless a foreign writer left a buffer
 * under-aligned for its type, in which case that column is copied (reader.rs:301 `align_buffers`).  The
 * validity view is dropped when null_count == 0 (reader.rs:271). */
AH_API ah_status ah_ipc_decode_batch(ah_context* ctx, const uint8_t* msg, int64_t msg_len, const void* body,
                                     int64_t body_len, int32_t n_fields, const ah_ipc_field* fields,
                                     ah_array_out* out_cols, int64_t* num_rows);
/* Header type (AH_IPC_*) and bodyLength of a framed message: what a stream reader needs before it fetches
 * the body. */
AH_API ah_status ah_ipc_message_info(ah_context* ctx, const uint8_t* msg, int64_t len, int32_t* header_type,
                                     int64_t* body_len);

/* The Arrow IPC FILE format (`FileWriter` arrow-ipc/src/writer.rs:1645-1768, `read_footer_length` / `FileDecoder`
 * arrow-ipc/src/reader.rs:944-1260) = "ARROW1" + padding, the same framed messages as the stream, an end-of-stream
 * marker, then the trailer these two functions build and parse: [Footer flatbuffer (File.fbs: version, schema,
 * dictionaries = [], recordBatches = [Block])][i32 footer length]["ARROW1"].  Host-only (ctx may be NULL).
 * ah_ipc_file_footer returns that trailer (free with ah_host_free).  ah_ipc_decode_footer takes the LAST tail_len
 * bytes of a file: with n_fields == NULL it only reports the footer length from the last 10 bytes (so the caller
 * knows how much to read); otherwise it needs the last footer_len + 10 bytes and returns the schema fields
 * (one block, ah_host_free) and the record-batch blocks (ah_host_free).  Errors carry the reference's text
 * ("Arrow file does not contain correct footer", "Invalid footer length: N"). */
AH_API ah_status ah_ipc_file_footer(ah_context* ctx, int32_t n_fields, const ah_ipc_field* fields, int32_t n_blocks,
                                    const ah_ipc_block* blocks, uint8_t** out, int64_t* out_len);
AH_API ah_status ah_ipc_decode_footer(ah_context* ctx, const uint8_t* tail, int64_t tail_len, int64_t* footer_len,
                                      int32_t* n_fields, ah_ipc_field** fields, int32_t* n_blocks,
                                      ah_ipc_block** blocks);
AH_API void ah_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* ARROW_HIP_H */
