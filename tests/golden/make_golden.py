"""Writes tests/golden/*.json: the inline golden vectors of the reference's own unit
tests for the hot path, TRANSCRIBED by hand from the cited Rust test bodies
(the reference is Rust-only and cannot run here; nothing below executes it).
Each case cites the reference test (file:line under /root/reference).

    python tests/golden/make_golden.py      # regenerates the JSON files
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
N = None
T, F = True, False


def arr(t, data, slice=None):
    d = {"type": t, "data": data}
    if slice:
        d["slice"] = slice
    return d


filter_cases = []
filter_cases.append(dict(name="doc_example", source="arrow-select/src/filter.rs:195-199",
                         values=arr("Int32", [5, 6, 7, 8, 9]), predicate=arr("Boolean", [T, F, F, T, F]),
                         expected=arr("Int32", [5, 8])))
for t in ["Date32", "Date64", "Time32(Second)", "Time32(Millisecond)", "Time64(Microsecond)",
          "Time64(Nanosecond)", "Duration(Second)", "Duration(Millisecond)", "Duration(Microsecond)",
          "Duration(Nanosecond)", "Timestamp(Second, None)", "Timestamp(Millisecond, None)",
          "Timestamp(Microsecond, None)", "Timestamp(Nanosecond, None)"]:
    filter_cases.append(dict(name=f"temporal_{t}", source="arrow-select/src/filter.rs:1090-1174",
                             values=arr(t, [1, 2, 3, 4]), predicate=arr("Boolean", [T, F, T, F]),
                             expected=arr(t, [1, 3])))
filter_cases.append(dict(name="test_filter_array_slice", source="arrow-select/src/filter.rs:1177",
                         values=arr("Int32", [5, 6, 7, 8, 9], [1, 4]), predicate=arr("Boolean", [T, F, F, T]),
                         expected=arr("Int32", [6, 9])))
# test_filter_array_low_density (:1191): 1..=65 then [66, 67]; mask true at value 65 and 67
vals = list(range(1, 66)) + [66, 67]
mask = [(v % 65 == 0) for v in range(1, 66)] + [F, T]
filter_cases.append(dict(name="test_filter_array_low_density", source="arrow-select/src/filter.rs:1191",
                         values=arr("Int32", vals), predicate=arr("Boolean", mask), expected=arr("Int32", [65, 67])))
# test_filter_array_high_density (:1207): [1, null, 3..=65] ++ [66, null, 67, null];
# mask [T x64, F] ++ [F, T, T, T]; expects len 67, null_count 3
vals = [1, N] + list(range(3, 66)) + [66, N, 67, N]
mask = [T] * 64 + [F] + [F, T, T, T]
exp = [1, N] + list(range(3, 65)) + [N, 67, N]
assert len(vals) == 69 and len(mask) == 69 and len(exp) == 67 and sum(x is None for x in exp) == 3
filter_cases.append(dict(name="test_filter_array_high_density", source="arrow-select/src/filter.rs:1207",
                         values=arr("Int32", vals), predicate=arr("Boolean", mask), expected=arr("Int32", exp),
                         expected_null_count=3))
filter_cases.append(dict(name="test_filter_primitive_array_with_null", source="arrow-select/src/filter.rs:1244",
                         values=arr("Int32", [5, N]), predicate=arr("Boolean", [F, T]), expected=arr("Int32", [N])))
filter_cases.append(dict(name="test_filter_array_slice_with_null", source="arrow-select/src/filter.rs:1414",
                         values=arr("Int32", [5, N, 7, 8, 9], [1, 4]), predicate=arr("Boolean", [T, F, F, T]),
                         expected=arr("Int32", [N, 9])))
filter_cases.append(dict(name="test_null_mask", source="arrow-select/src/filter.rs:1718",
                         values=arr("Int64", [1, 2, N]), predicate=arr("Boolean", [T, T, N]),
                         expected=arr("Int64", [1, 2])))
filter_cases.append(dict(name="test_fast_path_all_true", source="arrow-select/src/filter.rs:1738",
                         values=arr("Int64", [1, 2, N]), predicate=arr("Boolean", [T, T, T]),
                         expected=arr("Int64", [1, 2, N])))
filter_cases.append(dict(name="test_fast_path_all_false", source="arrow-select/src/filter.rs:1738",
                         values=arr("Int64", [1, 2, N]), predicate=arr("Boolean", [F, F, F]),
                         expected=arr("Int64", [])))
filter_cases.append(dict(name="filter_boolean", source="arrow-select/src/filter.rs:723-729 (filter_boolean; "
                         "values as in take.rs:1627)",
                         values=arr("Boolean", [F, N, T, F, N]), predicate=arr("Boolean", [T, T, T, F, T]),
                         expected=arr("Boolean", [F, N, T, N])))
filter_cases.append(dict(name="oversized_predicate", source="arrow-select/src/filter.rs:536-541",
                         values=arr("Int32", [1, 2, 3]), predicate=arr("Boolean", [T, F, T, T]),
                         error="InvalidArgumentError",
                         message="Filter predicate of length 4 is larger than target array of length 3"))
filter_cases.append(dict(name="test_filter_record_batch_no_columns", source="arrow-select/src/filter.rs:1727",
                         record_batch_rows=100, predicate=arr("Boolean", [T, T, N]), expected_rows=2))
# test_slices (:1758): 10 T, 30 F, 20 T, 17 F, 4 T
m = [T] * 10 + [F] * 30 + [T] * 20 + [F] * 17 + [T] * 4
filter_cases.append(dict(name="test_slices", source="arrow-select/src/filter.rs:1758", mask=m,
                         slices=[[0, 10], [40, 60], [77, 81]],
                         sliced=dict(offset=7, length=len(m) - 10, slices=[[0, 3], [33, 53], [70, 71]])))

filter_cases.append(dict(name="test_filter_string_array_simple", source="arrow-select/src/filter.rs:1233",
                         values=arr("Utf8", ["hello", " ", "world", "!"]), predicate=arr("Boolean", [T, F, T, F]),
                         expected=arr("Utf8", ["hello", "world"])))
filter_cases.append(dict(name="test_filter_string_array_with_null", source="arrow-select/src/filter.rs:1254",
                         values=arr("Utf8", ["hello", N, "world", N]), predicate=arr("Boolean", [T, F, F, T]),
                         expected=arr("Utf8", ["hello", N])))
filter_cases.append(dict(name="filter_large_string_with_null", source="arrow-select/src/filter.rs:1254 (LargeUtf8 arm :558)",
                         values=arr("LargeUtf8", ["hello", N, "world", N]), predicate=arr("Boolean", [T, F, F, T]),
                         expected=arr("LargeUtf8", ["hello", N])))

take_cases = []
for st in ["Utf8", "LargeUtf8"]:
    take_cases.append(dict(name=f"test_take_string_{st}", source="arrow-select/src/take.rs:1702-1733",
                           values=arr(st, ["one", N, "three", "four", "five"]), indices=arr("UInt32", [3, N, 1, 3, 4]),
                           expected=arr(st, ["four", N, N, "four", "five"])))
take_cases.append(dict(name="test_take_slice_string", source="arrow-select/src/take.rs:1736-1743",
                       values=arr("Utf8", ["hello", N, "world", N, "hi"]),
                       indices=arr("Int32", [0, 1, N, 0, 2], [1, 4]), expected=arr("Utf8", [N, N, "hello", "world"])))
take_cases.append(dict(name="non_null_indices", source="arrow-select/src/take.rs:1295",
                       values=arr("Int8", [N, 3, 5, 2, 3, N]), indices=arr("UInt32", [0, 5, 3, 1, 4, 2]),
                       expected=arr("Int8", [N, N, 2, 3, 3, 5])))
take_cases.append(dict(name="non_null_values", source="arrow-select/src/take.rs:1307",
                       values=arr("Int8", [0, 1, 2, 3, 4]), indices=arr("UInt32", [3, N, 1, 3, 2]),
                       expected=arr("Int8", [3, N, 1, 3, 2])))
take_cases.append(dict(name="non_null", source="arrow-select/src/take.rs:1319",
                       values=arr("Int8", [0, 3, 5, 2, 3, 1]), indices=arr("UInt32", [0, 5, 3, 1, 4, 2]),
                       expected=arr("Int8", [0, 1, 2, 3, 3, 5])))
take_cases.append(dict(name="nullable_indices_non_null_values_with_offset", source="arrow-select/src/take.rs:1331",
                       values=arr("Int64", [0, 10, 20, 30, 40, 50]),
                       indices=arr("UInt32", [0, 1, 2, 3, N, N], [2, 4]), expected=arr("Int64", [20, 30, N, N])))
take_cases.append(dict(name="nullable_indices_nullable_values_with_offset", source="arrow-select/src/take.rs:1351",
                       values=arr("Int64", [N, N, 20, 30, 40, 50]),
                       indices=arr("UInt32", [0, 1, 2, 3, N, N], [2, 4]), expected=arr("Int64", [20, 30, N, N])))
for t in ["Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64", "Float32", "Float64",
          "Date32", "Date64", "Time32(Second)", "Time64(Nanosecond)", "Duration(Second)",
          "Timestamp(Millisecond, None)"]:
    take_cases.append(dict(name=f"test_take_primitive_{t}", source="arrow-select/src/take.rs:1371-1551",
                           values=arr(t, [0, N, 2, 3, N]), indices=arr("UInt32", [3, N, 1, 3, 2]),
                           expected=arr(t, [3, N, N, 3, 2])))
for it in ["Int64", "Int16", "UInt64", "UInt8", "Int8", "UInt16", "Int32"]:
    take_cases.append(dict(name=f"index_type_{it}", source="arrow-select/src/take.rs:1553-1625",
                           values=arr("Int64", [0, N, 2, -15, N]), indices=arr(it, [3, N, 1, 3, 2]),
                           expected=arr("Int64", [-15, N, N, -15, 2])))
take_cases.append(dict(name="index_int64_f32", source="arrow-select/src/take.rs:1553-1596",
                       values=arr("Float32", [0.0, N, 2.21, -3.1, N]), indices=arr("Int64", [3, N, 1, 3, 2]),
                       expected=arr("Float32", [-3.1, N, N, -3.1, 2.21])))
take_cases.append(dict(name="test_take_bool", source="arrow-select/src/take.rs:1627",
                       values=arr("Boolean", [F, N, T, F, N]), indices=arr("UInt32", [3, N, 1, 3, 2]),
                       expected=arr("Boolean", [F, N, N, F, T])))
take_cases.append(dict(name="test_take_bool_nullable_index", source="arrow-select/src/take.rs:1639",
                       values=arr("Boolean", [T, N, F]),
                       indices=dict(type="UInt32", raw=[99, 0, 999, 1, 9999, 2], valid=[F, T, F, T, F, T]),
                       expected=arr("Boolean", [N, T, N, N, N, F])))
take_cases.append(dict(name="test_take_bool_nullable_index_nonnull_values", source="arrow-select/src/take.rs:1662",
                       values=arr("Boolean", [T, T, F]),
                       indices=dict(type="UInt32", raw=[99, 0, 999, 1, 9999, 2], valid=[F, T, F, T, F, T]),
                       expected=arr("Boolean", [N, T, N, T, N, F])))
# round 6: the remaining in-scope inline vectors of take.rs (sliced indices over Boolean values, sliced string values, null
# indices whose slots hold out-of-range numbers over a string column)
take_cases.append(dict(name="test_take_bool_with_offset", source="arrow-select/src/take.rs:1685-1700",
                       values=arr("Boolean", [F, N, T, F, N]), indices=arr("UInt32", [3, N, 1, 3, 2, N], [2, 4]),
                       expected=arr("Boolean", [N, F, T, N])))
SL7 = ["aaa", "bbb", N, "ccccc", "dd", N, "eeee"]
take_cases.append(dict(name="test_take_bytes_sliced_values_all_valid", source="arrow-select/src/take.rs:1750-1770",
                       values=arr("Utf8", SL7, [2, 5]), indices=arr("Int32", [1, 2, 4, 1]),
                       expected=arr("Utf8", ["ccccc", "dd", "eeee", "ccccc"])))
take_cases.append(dict(name="test_take_bytes_sliced_values_nullable", source="arrow-select/src/take.rs:1772-1778",
                       values=arr("Utf8", SL7, [2, 5]), indices=arr("Int32", [1, N, 0, 4, 3]),
                       expected=arr("Utf8", ["ccccc", N, N, "eeee", N])))
take_cases.append(dict(name="test_take_bytes_null_indices", source="arrow-select/src/take.rs:2719-2728",
                       values=arr("Utf8", ["foo", N]), indices=dict(type="Int32", raw=[0, 1, 400, 400], valid=[T, T, F, F]),
                       expected=arr("Utf8", ["foo", N, N, N])))
take_cases.append(dict(name="test_take_null_indices", source="arrow-select/src/take.rs:2686-2699",
                       values=arr("Int32", [1, 23, 4, 5]),
                       indices=dict(type="Int32", raw=[1, 2, 400, 400], valid=[T, T, F, F]),
                       expected=arr("Int32", [23, 4, N, N])))
take_cases.append(dict(name="test_take_out_of_bounds", source="arrow-select/src/take.rs:2408-2420",
                       values=arr("Int64", [0, N, 2, 3, N]), indices=arr("UInt32", [3, N, 1, 3, 6]),
                       check_bounds=True, error="ComputeError",
                       message="Array index out of bounds, cannot get item at index 6 from 5 entries"))
take_cases.append(dict(name="test_take_out_of_bounds_panic", source="arrow-select/src/take.rs:2423-2434",
                       values=arr("Int64", [0, 1, 2, 3]), indices=arr("UInt32", [1000]),
                       panic="index out of bounds: the len is 4 but the index is 1000"))
take_cases.append(dict(name="check_bounds_message", source="arrow-select/src/take.rs:2457-2466 (NullArray(5) "
                       "replaced by an Int32 array of 5 rows: the message depends only on len)",
                       values=arr("Int32", [0, 0, 0, 0, 0]), indices=arr("UInt32", [0, N, 15]), check_bounds=True,
                       error="ComputeError",
                       message="Array index out of bounds, cannot get item at index 15 from 5 entries"))
take_cases.append(dict(name="empty_indices", source="arrow-select/src/take.rs:215-217",
                       values=arr("Int32", [1, 2, 3]), indices=arr("UInt32", []), expected=arr("Int32", [])))

# test_take_decimal128_non_null_indices / test_take_decimal128 (take.rs:1263-1293): 16-byte natives
take_cases.append(dict(name="test_take_decimal128_non_null_indices", source="arrow-select/src/take.rs:1263-1276",
                       values=arr("Decimal128(10, 5)", [N, 3, 5, 2, 3, N]), indices=arr("UInt32", [0, 5, 3, 1, 4, 2]),
                       expected=arr("Decimal128(10, 5)", [N, N, 2, 3, 3, 5])))
take_cases.append(dict(name="test_take_decimal128", source="arrow-select/src/take.rs:1279-1292",
                       values=arr("Decimal128(10, 5)", [0, 1, 2, 3, 4]), indices=arr("UInt32", [3, N, 1, 3, 2]),
                       expected=arr("Decimal128(10, 5)", [3, N, 1, 3, 2])))

arith_cases = []
a, b = [4, 3, 5, -6, 100], [6, 2, 5, -7, 3]
for op, exp in [("add", [10, 5, 10, -13, 103]), ("sub", [-2, 1, 0, 1, 97]), ("div", [0, 1, 1, 0, 33]),
                ("mul", [24, 6, 25, 42, 300]), ("rem", [4, 1, 0, -6, 1])]:
    arith_cases.append(dict(name=f"test_integer_{op}", source="arrow-arith/src/numeric.rs:1296-1316", op=op,
                            lhs=arr("Int32", a), rhs=arr("Int32", b), expected=arr("Int32", exp)))
arith_cases.append(dict(name="test_integer_nulls", source="arrow-arith/src/numeric.rs:1318-1322", op="add",
                        lhs=arr("Int8", [2, N, 45]), rhs=arr("Int8", [5, 3, N]), expected=arr("Int8", [7, N, N])))
arith_cases.append(dict(name="u8_add_overflow", source="arrow-arith/src/numeric.rs:1324-1330", op="add",
                        lhs=arr("UInt8", [56, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), error="ArithmeticOverflow",
                        message="Overflow happened on: 56 + 200",
                        display="Arithmetic overflow: Overflow happened on: 56 + 200"))
arith_cases.append(dict(name="u8_add_wrapping", source="arrow-arith/src/numeric.rs:1332-1333", op="add_wrapping",
                        lhs=arr("UInt8", [56, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), expected=arr("UInt8", [0, 7, 8])))
arith_cases.append(dict(name="u8_sub_overflow", source="arrow-arith/src/numeric.rs:1335-1339", op="sub",
                        lhs=arr("UInt8", [34, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), error="ArithmeticOverflow",
                        message="Overflow happened on: 34 - 200"))
arith_cases.append(dict(name="u8_sub_wrapping", source="arrow-arith/src/numeric.rs:1341-1342", op="sub_wrapping",
                        lhs=arr("UInt8", [34, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), expected=arr("UInt8", [90, 3, 254])))
arith_cases.append(dict(name="u8_mul_overflow", source="arrow-arith/src/numeric.rs:1344-1348", op="mul",
                        lhs=arr("UInt8", [34, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), error="ArithmeticOverflow",
                        message="Overflow happened on: 34 * 200"))
arith_cases.append(dict(name="u8_mul_wrapping", source="arrow-arith/src/numeric.rs:1350-1351", op="mul_wrapping",
                        lhs=arr("UInt8", [34, 5, 3]), rhs=arr("UInt8", [200, 2, 5]), expected=arr("UInt8", [144, 10, 15])))
arith_cases.append(dict(name="i16_min_div_neg1", source="arrow-arith/src/numeric.rs:1353-1358", op="div",
                        lhs=arr("Int16", [-32768]), rhs=arr("Int16", [-1]), error="ArithmeticOverflow",
                        message="Overflow happened on: -32768 / -1"))
arith_cases.append(dict(name="i16_min_rem_neg1", source="arrow-arith/src/numeric.rs:1347-1350", op="rem",
                        lhs=arr("Int16", [-32768]), rhs=arr("Int16", [-1]), expected=arr("Int16", [0])))
arith_cases.append(dict(name="i16_div_zero", source="arrow-arith/src/numeric.rs:1353-1361", op="div",
                        lhs=arr("Int16", [21]), rhs=arr("Int16", [0]), error="DivideByZero",
                        message="Divide by zero error", display="Divide by zero error"))
arith_cases.append(dict(name="i16_rem_zero", source="arrow-arith/src/numeric.rs:1353-1361", op="rem",
                        lhs=arr("Int16", [21]), rhs=arr("Int16", [0]), error="DivideByZero",
                        message="Divide by zero error"))
FMAX = 3.4028234663852886e38
fa = [1.0, FMAX, 6.0, -4.0, -1.0, 0.0]
fb = [1.0, FMAX, FMAX, -3.0, 45.0, 0.0]
arith_cases.append(dict(name="test_float_add", source="arrow-arith/src/numeric.rs:1364-1372", op="add",
                        lhs=arr("Float32", fa), rhs=arr("Float32", fb),
                        expected=arr("Float32", [2.0, "inf", FMAX, -7.0, 44.0, 0.0])))
arith_cases.append(dict(name="test_float_sub", source="arrow-arith/src/numeric.rs:1374-1377", op="sub",
                        lhs=arr("Float32", fa), rhs=arr("Float32", fb),
                        expected=arr("Float32", [0.0, 0.0, -FMAX, -1.0, -46.0, 0.0])))
arith_cases.append(dict(name="test_float_mul", source="arrow-arith/src/numeric.rs:1379-1382", op="mul",
                        lhs=arr("Float32", fa), rhs=arr("Float32", fb),
                        expected=arr("Float32", [1.0, "inf", "inf", 12.0, -45.0, 0.0])))
arith_cases.append(dict(name="test_float_rem", source="arrow-arith/src/numeric.rs:1391-1397", op="rem",
                        lhs=arr("Float32", fa), rhs=arr("Float32", fb),
                        expected=arr("Float32", [0.0, 0.0, 6.0, -1.0, -1.0, "nan"])))
arith_cases.append(dict(name="type_mismatch", source="arrow-arith/src/numeric.rs:270-272", op="add",
                        lhs=arr("Int32", [1]), rhs=arr("Int64", [1]), error="InvalidArgumentError",
                        message="Invalid arithmetic operation: Int32 + Int64"))

# arity.rs tests (arrow-arith/src/arity.rs:460-540): the closures are `l + r`, i.e. add / add_wrapping without overflow
for opn in ["add", "add_wrapping"]:
    arith_cases.append(dict(name=f"test_binary_mut_{opn}", source="arrow-arith/src/arity.rs:460-467 (test_binary_mut), "
                            ":489-496 (test_try_binary_mut)", op=opn, lhs=arr("Int32", [15, 14, 9, 8, 1]),
                            rhs=arr("Int32", [1, N, 3, N, 5]), expected=arr("Int32", [16, N, 12, N, 6])))
    arith_cases.append(dict(name=f"test_try_binary_mut_no_nulls_{opn}", source="arrow-arith/src/arity.rs:498-502", op=opn,
                            lhs=arr("Int32", [15, 14, 9, 8, 1]), rhs=arr("Int32", [1, 2, 3, 4, 5]),
                            expected=arr("Int32", [16, 16, 12, 12, 6])))
    arith_cases.append(dict(name=f"test_binary_mut_null_buffer_{opn}", source="arrow-arith/src/arity.rs:470-486", op=opn,
                            lhs=arr("Int32", [3, 4, 5, 6, N]),
                            rhs=dict(type="Int32", raw=[10, 11, 12, 13, 14], valid=[T, T, T, T, T]),
                            expected=arr("Int32", [13, 15, 17, 19, N])))
    arith_cases.append(dict(name=f"test_try_binary_mut_all_valid_null_buffers_{opn}", source="arrow-arith/src/arity.rs:521-531",
                            op=opn, lhs=dict(type="Int32", raw=[1, 2], valid=[T, T]),
                            rhs=dict(type="Int32", raw=[10, 20], valid=[T, T]), expected=arr("Int32", [11, 22])))

cmp_cases = []
eights = [8] * 10
seq = [6, 7, 8, 9, 10, 6, 7, 8, 9, 10]
cmp_cases.append(dict(name="i64_lt", source="arrow-ord/src/comparison.rs:526", op="lt",
                      lhs=arr("Int64", eights), rhs=arr("Int64", seq),
                      expected=arr("Boolean", [F, F, F, T, T, F, F, F, T, T])))
cmp_cases.append(dict(name="i64_lt_eq", source="arrow-ord/src/comparison.rs:650", op="lt_eq",
                      lhs=arr("Int64", eights), rhs=arr("Int64", seq),
                      expected=arr("Boolean", [F, F, T, T, T, F, F, T, T, T])))
cmp_cases.append(dict(name="i64_lt_scalar", source="arrow-ord/src/comparison.rs:612", op="lt",
                      lhs=arr("Int64", seq), rhs_scalar=dict(type="Int64", value=8),
                      expected=arr("Boolean", [T, T, F, F, F, T, T, F, F, F])))
cmp_cases.append(dict(name="i64_lt_nulls", source="arrow-ord/src/comparison.rs:622", op="lt",
                      lhs=arr("Int64", [N, N, 1, 1, N, N, 2, 2]), rhs=arr("Int64", [N, 1, N, 1, N, 3, N, 3]),
                      expected=arr("Boolean", [N, N, N, F, N, N, N, T])))
cmp_cases.append(dict(name="i64_lt_scalar_nulls", source="arrow-ord/src/comparison.rs:640", op="lt",
                      lhs=arr("Int64", [N, 1, 2, 3, N, 1, 2, 3, 2, N]), rhs_scalar=dict(type="Int64", value=2),
                      expected=arr("Boolean", [N, T, F, F, N, T, F, F, F, N])))
fl = ["nan", 7.0, 8.0, 8.0, 11.0, "nan"]
fr = ["nan", "nan", 8.0, 9.0, 10.0, 1.0]
cmp_cases.append(dict(name="f64_lt_total_order", source="arrow-ord/src/comparison.rs:2525-2534", op="lt",
                      lhs=arr("Float64", fl), rhs=arr("Float64", fr), expected=arr("Boolean", [F, T, F, T, F, F])))
cmp_cases.append(dict(name="f64_lt_eq_total_order", source="arrow-ord/src/comparison.rs:2525-2534", op="lt_eq",
                      lhs=arr("Float64", fl), rhs=arr("Float64", fr), expected=arr("Boolean", [T, T, T, T, F, F])))
cmp_cases.append(dict(name="f32_lt_total_order", source="arrow-ord/src/comparison.rs:2525-2534", op="lt",
                      lhs=arr("Float32", fl), rhs=arr("Float32", fr), expected=arr("Boolean", [F, T, F, T, F, F])))
el = ["nan", 7.0, 8.0, 8.0, 10.0]
er = ["nan", "nan", 8.0, 8.0, 10.0]
cmp_cases.append(dict(name="f64_eq", source="arrow-ord/src/comparison.rs:2494-2501", op="eq",
                      lhs=arr("Float64", el), rhs=arr("Float64", er), expected=arr("Boolean", [T, F, T, T, T])))
cmp_cases.append(dict(name="f64_neq", source="arrow-ord/src/comparison.rs:2494-2501", op="neq",
                      lhs=arr("Float64", el), rhs=arr("Float64", er), expected=arr("Boolean", [F, T, F, F, F])))
# the reference only asserts the LENGTH of the result buffer (130 is not a multiple of 64); 1 >= 2 is false
cmp_cases.append(dict(name="test_length_of_result_buffer", source="arrow-ord/src/comparison.rs:791-803", op="gt_eq",
                      lhs=arr("Int8", [1] * 130), rhs=arr("Int8", [2] * 130), expected=arr("Boolean", [F] * 130)))
cmp_cases.append(dict(name="length_mismatch", source="arrow-ord/src/cmp.rs:228-232", op="eq",
                      lhs=arr("Int32", [1, 2]), rhs=arr("Int32", [1]), error="InvalidArgumentError",
                      message="Cannot compare arrays of different lengths, got 2 vs 1"))
cmp_cases.append(dict(name="type_mismatch", source="arrow-ord/src/cmp.rs:260-264", op="lt",
                      lhs=arr("Int32", [1]), rhs=arr("Int64", [1]), error="InvalidArgumentError",
                      message="Invalid comparison operation: Int32 < Int64"))

# --- the cmp_i64! / cmp_i64_scalar! matrix (comparison.rs:201-790; each case is also run x10-replicated by the tests)
def cmp_case(name, line, op, lhs, rhs=None, expected=None, t="Int64", **kw):
    d = dict(name=name, source=f"arrow-ord/src/comparison.rs:{line}", op=op, lhs=arr(t, lhs) if isinstance(lhs, list) else lhs,
             expected=arr("Boolean", expected))
    if rhs is not None:
        d["rhs"] = arr(t, rhs) if isinstance(rhs, list) else rhs
    d.update(kw)
    cmp_cases.append(d)


def sc(t, v):
    return dict(type=t, value=v)


cmp_case("test_primitive_array_eq", "201-210", "eq", eights, seq, [F, F, T, F, F, F, F, T, F, F])
for tt in ["Timestamp(Second, None)", "Time32(Second)", "Time32(Millisecond)", "Time64(Microsecond)", "Time64(Nanosecond)"]:
    cmp_case(f"test_primitive_array_eq_{tt}", "211-250", "eq", eights, seq, [F, F, T, F, F, F, F, T, F, F], t=tt)
cmp_case("test_primitive_array_eq_scalar", "291-298", "eq", seq, expected=[F, F, T, F, F, F, F, T, F, F], rhs_scalar=sc("Int64", 8))
cmp_case("test_primitive_array_eq_with_slice", "301-311", "eq", arr("Int32", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [5, 5]),
         arr("Int32", [6, 7, 8, 8, 10]), [T, T, T, F, T])
cmp_case("test_primitive_array_eq_scalar_with_slice", "314-322", "eq", arr("Int32", [1, N, 2, 3], [1, 3]),
         expected=[N, T, F], rhs_scalar=sc("Int32", 2))
cmp_case("test_primitive_array_neq", "325-332", "neq", eights, seq, [T, T, F, T, T, T, T, F, T, T])
cmp_case("test_primitive_array_neq_scalar", "343-350", "neq", seq, expected=[T, T, F, T, T, T, T, F, T, T], rhs_scalar=sc("Int64", 8))
# round 6: the rest of comparison.rs' scalar-operand tests (lt against a scalar, with nulls; the *_dyn_scalar family over
# Int32 / Float32 / Float64; the temporal arrays of test_primitive_dyn_scalar; Boolean and Utf8 scalars; signed zeros)
cmp_case("test_primitive_array_lt_scalar", "611-618", "lt", [6, 7, 8, 9, 10, 6, 7, 8, 9, 10], expected=[T, T, F, F, F, T, T, F, F, F],
         rhs_scalar=sc("Int64", 8))
LTN_A, LTN_B = [N, N, 1, 1, N, N, 2, 2], [N, 1, N, 1, N, 3, N, 3]
cmp_case("test_primitive_array_lt_nulls", "621-636", "lt", LTN_A, LTN_B, [N, N, N, F, N, N, N, T])
cmp_case("test_primitive_array_lt_nulls_Timestamp(Millisecond, None)", "621-636", "lt", LTN_A, LTN_B, [N, N, N, F, N, N, N, T],
         t="Timestamp(Millisecond, None)")
cmp_case("test_primitive_array_lt_scalar_nulls", "639-646", "lt", [N, 1, 2, 3, N, 1, 2, 3, 2, N],
         expected=[N, T, F, F, N, T, F, F, F, N], rhs_scalar=sc("Int64", 2))
DYN = [6, 7, 8, 8, 10]
for op, line_i, line_f, e in [("eq", "1448-1457", "1475-1485", [F, F, T, T, F]), ("lt", "1488-1493", "1511-1521", [T, T, F, F, F]),
                              ("lt_eq", "1524-1529", "1664-1673", [T, T, T, T, F]), ("gt", "1676-1686", "1704-1713", [F, F, F, F, T]),
                              ("gt_eq", "1716-1721", "1739-1748", [F, F, T, T, T]), ("neq", "1751-1760", "1777-1787", [T, T, F, F, T])]:
    cmp_case(f"test_{op}_dyn_scalar", line_i, op, DYN, expected=e, t="Int32", rhs_scalar=sc("Int32", 8))
    for ft in ("Float32", "Float64"):
        cmp_case(f"test_{op}_dyn_scalar_float_{ft}", line_f, op, [6.0, 7.0, 8.0, 8.0, 10.0], expected=e, t=ft, rhs_scalar=sc(ft, 8.0))
PD = [1, N, 8, N, 10]  # test_primitive_dyn_scalar (:1532-1563) against the scalar 8
for tt, line in [("Date32", "1580-1584"), ("Date64", "1586-1590"), ("Time32(Second)", "1592-1599"), ("Time32(Millisecond)", "1592-1599"),
                 ("Time64(Microsecond)", "1601-1608"), ("Time64(Nanosecond)", "1601-1608"), ("Duration(Second)", "1634-1644"),
                 ("Duration(Millisecond)", "1634-1644"), ("Duration(Microsecond)", "1634-1644"), ("Duration(Nanosecond)", "1634-1644"),
                 ("Timestamp(Second, None)", "1566-1577"), ("Timestamp(Microsecond, None)", "1566-1577"),
                 ("Timestamp(Nanosecond, None)", "1566-1577")]:
    for op, e in [("eq", [F, N, T, N, F]), ("gt_eq", [F, N, T, N, T]), ("gt", [F, N, F, N, T]), ("lt_eq", [T, N, T, N, F]), ("lt", [T, N, F, N, F])]:
        cmp_case(f"test_primitive_dyn_scalar_{tt}_{op}", line, op, PD, expected=e, t=tt, rhs_scalar=sc(tt, 8))
for op, line, a, e in [("eq", "2077-2085", [T, F, T], [F, T, F]), ("lt", "2088-2096", [T, F, T, N], [F, F, F, N]),
                       ("gt", "2099-2107", [T, F, T], [T, F, T]), ("lt_eq", "2110-2118", [T, F, T], [F, T, F]),
                       ("gt_eq", "2121-2129", [T, F, T], [T, T, T]), ("neq", "2132-2140", [T, F, T], [T, F, T])]:
    cmp_case(f"test_{op}_dyn_bool_scalar", line, op, a, expected=e, t="Boolean", rhs_scalar=sc("Boolean", F))
cmp_case("test_floating_zeros", "3557-3561", "eq", [0.0, -0.0], [-0.0, 0.0], [F, F], t="Float32")
cmp_case("test_floating_zeros_scalar_pos", "3564-3567", "eq", [0.0, -0.0], expected=[T, F], t="Float32", rhs_scalar=sc("Float32", 0.0))
cmp_case("test_floating_zeros_scalar_neg", "3569-3572", "eq", [0.0, -0.0], expected=[F, T], t="Float32", rhs_scalar=sc("Float32", -0.0))
BA, BB = [T, F, F, T, T, N], [T, T, F, F, N, F]
for op, line, e in [("eq", "353-366", [T, F, T, F, N, N]), ("neq", "368-381", [F, T, F, T, N, N]),
                    ("lt", "383-396", [F, T, F, F, N, N]), ("lt_eq", "398-411", [T, T, T, F, N, N]),
                    ("gt", "413-426", [F, F, F, T, N, N]), ("gt_eq", "428-441", [T, F, T, T, N, N])]:
    cmp_case(f"test_boolean_array_{op}", line, op, BA, BB, e, t="Boolean")
for op, line, ef, et in [("eq", "443-453", [F, T, N], [T, F, N]), ("neq", "456-467", [T, F, N], [F, T, N]),
                         ("lt", "470-482", [F, F, N], [F, T, N]), ("lt_eq", "485-496", [F, T, N], [T, T, N]),
                         ("gt", "499-509", [T, F, N], [F, F, N]), ("gt_eq", "512-523", [T, T, N], [T, F, N])]:
    cmp_case(f"test_boolean_array_{op}_scalar_false", line, op, [T, F, N], expected=ef, t="Boolean", rhs_scalar=sc("Boolean", F))
    cmp_case(f"test_boolean_array_{op}_scalar_true", line, op, [T, F, N], expected=et, t="Boolean", rhs_scalar=sc("Boolean", T))
cmp_case("test_primitive_array_lt_eq_scalar", "660-667", "lt_eq", seq, expected=[T, T, T, F, F, T, T, T, F, F], rhs_scalar=sc("Int64", 8))
cmp_case("test_primitive_array_lt_eq_nulls", "670-677", "lt_eq", [N, N, 1, N, N, 1, N, N, 1], [N, 1, 0, N, 1, 2, N, N, 3],
         [N, N, F, N, N, T, N, N, T])
cmp_case("test_primitive_array_lt_eq_scalar_nulls", "680-687", "lt_eq", [N, 1, 2, N, 1, 2, N, 1, 2],
         expected=[N, T, F, N, T, F, N, T, F], rhs_scalar=sc("Int64", 1))
cmp_case("test_primitive_array_gt", "690-697", "gt", eights, seq, [T, T, F, F, F, T, T, F, F, F])
cmp_case("test_primitive_array_gt_scalar", "700-707", "gt", seq, expected=[F, F, F, T, T, F, F, F, T, T], rhs_scalar=sc("Int64", 8))
cmp_case("test_primitive_array_gt_nulls", "710-717", "gt", [N, N, 1, N, N, 2, N, N, 3], [N, 1, 1, N, 1, 1, N, 1, 1],
         [N, N, F, N, N, T, N, N, T])
cmp_case("test_primitive_array_gt_scalar_nulls", "720-727", "gt", [N, 1, 2, N, 1, 2, N, 1, 2],
         expected=[N, F, T, N, F, T, N, F, T], rhs_scalar=sc("Int64", 1))
cmp_case("test_primitive_array_gt_eq", "730-737", "gt_eq", eights, seq, [T, T, T, F, F, T, T, T, F, F])
cmp_case("test_primitive_array_gt_eq_scalar", "740-747", "gt_eq", seq, expected=[F, F, T, T, T, F, F, T, T, T], rhs_scalar=sc("Int64", 8))
cmp_case("test_primitive_array_gt_eq_nulls", "750-757", "gt_eq", [N, N, 1, N, 1, 2, N, N, 1], [N, 1, N, N, 1, 1, N, 2, 2],
         [N, N, N, N, T, T, N, N, F])
cmp_case("test_primitive_array_gt_eq_scalar_nulls", "760-767", "gt_eq", [N, 1, 2, N, 2, 3, N, 3, 4],
         expected=[N, F, T, N, T, T, N, T, T], rhs_scalar=sc("Int64", 2))
cmp_case("test_primitive_array_compare_slice", "770-778", "lt", arr("Int32", list(range(100)), [50, 50]),
         arr("Int32", list(range(100, 200)), [50, 50]), [T] * 50)
cmp_case("test_primitive_array_compare_scalar_slice", "781-788", "lt", arr("Int32", list(range(100)), [50, 50]),
         expected=[T] * 50, rhs_scalar=sc("Int32", 200))
# --- NaN under the total order, gt / gt_eq and NaN scalars (comparison.rs:2538-2658)
for ft in ["Float32", "Float64"]:
    cmp_case(f"{ft}_gt_total_order", "2538-2566", "gt", fl, fr, [F, F, F, F, T, T], t=ft)
    cmp_case(f"{ft}_gt_eq_total_order", "2538-2566", "gt_eq", fl, fr, [T, F, T, F, T, T], t=ft)
    nan_l = ["nan", 7.0, 8.0, 8.0, 10.0]
    for op, line, e in [("eq", "2569-2595", [T, F, F, F, F]), ("neq", "2569-2595", [F, T, T, T, T]),
                        ("lt", "2598-2625", [F, T, T, T, T]), ("lt_eq", "2598-2625", [T, T, T, T, T]),
                        ("gt", "2628-2658", [F, F, F, F, F]), ("gt_eq", "2628-2658", [T, F, F, F, F])]:
        cmp_case(f"{ft}_{op}_nan_scalar", line, op, nan_l, expected=e, t=ft, rhs_scalar=sc(ft, "nan"))
cmp_case("f32_eq", "2475-2503", "eq", el, er, [T, F, T, T, T], t="Float32")
cmp_case("f32_neq", "2475-2503", "neq", el, er, [F, T, F, F, F], t="Float32")
cmp_case("f32_lt_eq_total_order", "2506-2535", "lt_eq", fl, fr, [T, T, T, T, F, F], t="Float32")


# --- arrow-ord/src/cmp.rs tests: distinct / not_distinct and scalar-vs-scalar (:1028-1170)
def cmp_rs(name, line, op, expected, **kw):
    d = dict(name=name, source=f"arrow-ord/src/cmp.rs:{line}", op=op, expected=arr("Boolean", expected))
    d.update(kw)
    cmp_cases.append(d)


cmp_rs("is_distinct_from_non_nulls", "1028-1040", "distinct", [T, T, F, T, T], lhs=arr("Int32", [0, 1, 2, 3, 4]), rhs=arr("Int32", [4, 3, 2, 1, 0]))
cmp_rs("is_not_distinct_from_non_nulls", "1028-1040", "not_distinct", [F, F, T, F, F], lhs=arr("Int32", [0, 1, 2, 3, 4]),
       rhs=arr("Int32", [4, 3, 2, 1, 0]))
DL = dict(type="Int32", raw=[0, 0, 1, 3, 0, 0], valid=[T, T, F, T, T, T])   # values under the null slot are NOT zero
DR = dict(type="Int32", raw=[0] * 6, valid=[T, F, F, F, T, F])
cmp_rs("is_distinct_from_nulls", "1043-1066", "distinct", [F, T, F, T, F, T], lhs=DL, rhs=DR)
cmp_rs("is_not_distinct_from_nulls", "1043-1066", "not_distinct", [T, F, T, F, T, F], lhs=DL, rhs=DR)
S12, SNULL = sc("Int32", 12), sc("Int32", N)
cmp_rs("test_distinct_scalar_12_12", "1069-1074", "distinct", [F], lhs_scalar=S12, rhs_scalar=S12)
cmp_rs("test_not_distinct_scalar_12_12", "1069-1074", "not_distinct", [T], lhs_scalar=S12, rhs_scalar=S12)
# a scalar against a length-1 null ARRAY, both orders (:1076-1082)
cmp_rs("test_distinct_scalar_vs_null_array", "1076-1082", "distinct", [T], lhs_scalar=S12, rhs=arr("Int32", [N]))
cmp_rs("test_not_distinct_scalar_vs_null_array", "1076-1082", "not_distinct", [F], lhs_scalar=S12, rhs=arr("Int32", [N]))
cmp_rs("test_distinct_null_array_vs_scalar", "1076-1082", "distinct", [T], lhs=arr("Int32", [N]), rhs_scalar=S12)
cmp_rs("test_not_distinct_null_array_vs_scalar", "1076-1082", "not_distinct", [F], lhs=arr("Int32", [N]), rhs_scalar=S12)
cmp_rs("test_distinct_scalar_vs_null_scalar", "1084-1086", "distinct", [T], lhs_scalar=S12, rhs_scalar=SNULL)
cmp_rs("test_not_distinct_scalar_vs_null_scalar", "1084-1086", "not_distinct", [F], lhs_scalar=S12, rhs_scalar=SNULL)
cmp_rs("test_distinct_null_scalars", "1088-1089", "distinct", [F], lhs_scalar=SNULL, rhs_scalar=SNULL)
cmp_rs("test_not_distinct_null_scalars", "1088-1089", "not_distinct", [T], lhs_scalar=SNULL, rhs_scalar=SNULL)
DA = dict(type="Int32", raw=[0, 1, 2, 3], valid=[F, F, T, T])
for sname, sv, lines, e_d, e_nd in [("null", N, "1091-1102", [F, F, T, T], [T, T, F, F]),
                                    ("1", 1, "1104-1110", [T, T, T, T], [F, F, F, F]),
                                    ("3", 3, "1112-1118", [T, T, T, F], [F, F, F, T])]:
    cmp_rs(f"test_distinct_array_vs_scalar_{sname}", lines, "distinct", e_d, lhs=DA, rhs_scalar=sc("Int32", sv))
    cmp_rs(f"test_distinct_scalar_{sname}_vs_array", lines, "distinct", e_d, lhs_scalar=sc("Int32", sv), rhs=DA)
    cmp_rs(f"test_not_distinct_array_vs_scalar_{sname}", lines, "not_distinct", e_nd, lhs=DA, rhs_scalar=sc("Int32", sv))
    cmp_rs(f"test_not_distinct_scalar_{sname}_vs_array", lines, "not_distinct", e_nd, lhs_scalar=sc("Int32", sv), rhs=DA)
S54 = sc("Int32", 54)
cmp_rs("test_scalar_negation_eq", "1151-1160", "eq", [T], lhs_scalar=S54, rhs_scalar=S54)
cmp_rs("test_scalar_negation_neq", "1151-1160", "neq", [F], lhs_scalar=S54, rhs_scalar=S54)
cmp_rs("test_scalar_empty_array_vs_scalar", "1162-1170", "eq", [], lhs=arr("Int32", []), rhs_scalar=sc("Int32", 23))
cmp_rs("test_scalar_empty_scalar_vs_array", "1162-1170", "eq", [], lhs_scalar=sc("Int32", 23), rhs=arr("Int32", []))

cast_cases = []
I64MIN, I64MAX = -9223372036854775808, 9223372036854775807
cast_cases.append(dict(name="test_cast_from_int64_to_f64", source="arrow-cast/src/cast/mod.rs:8449-8480",
                       values=arr("Int64", [I64MIN, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, I64MAX]),
                       to="Float64",
                       expected=arr("Float64", [-9223372036854775808.0, -2147483648.0, -32768.0, -128.0, 0.0, 127.0,
                                                32767.0, 2147483647.0, 9223372036854775808.0])))
cast_cases.append(dict(name="test_cast_from_int64_to_i32", source="arrow-cast/src/cast/mod.rs:8449-8480",
                       values=arr("Int64", [I64MIN, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, I64MAX]),
                       to="Int32",
                       expected=arr("Int32", [N, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, N])))
cast_cases.append(dict(name="test_cast_from_int64_to_u8", source="arrow-cast/src/cast/mod.rs:8449-8480",
                       values=arr("Int64", [I64MIN, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, I64MAX]),
                       to="UInt8", expected=arr("UInt8", [N, N, N, N, 0, 127, N, N, N])))
cast_cases.append(dict(name="f64_to_utf8_pinned", source="arrow-cast/src/cast/mod.rs:4857-4876; "
                       "arrow-cast/src/pretty.rs:865-899; arrow-csv/src/writer.rs:705",
                       values=arr("Float64", [1.5, 2.5, N, 3.2234, 123.564532, -556132.25]), to="Utf8",
                       expected=arr("Utf8", ["1.5", "2.5", N, "3.2234", "123.564532", "-556132.25"])))
# two more Float64 texts the reference holds, through the same ArrayFormatter (arrow-csv/src/writer.rs builds its cells with it)
cast_cases.append(dict(name="f64_to_utf8_csv_writer_quote_style", source="arrow-csv/src/writer.rs:1374-1385 (test_write_csv_quote_style: "
                       "float column 1.1, 2.2, 3.3, 4.4)", values=arr("Float64", [1.1, 2.2, 3.3, 4.4]), to="Utf8",
                       expected=arr("Utf8", ["1.1", "2.2", "3.3", "4.4"])))
cast_cases.append(dict(name="f64_to_utf8_csv_writer_doc_example", source="arrow-csv/src/writer.rs:144-163 (module doc test: price "
                       "1.50, 2.25, 3.00)", values=arr("Float64", [1.50, 2.25, 3.00]), to="LargeUtf8",
                       expected=arr("LargeUtf8", ["1.5", "2.25", "3.0"])))
cast_cases.append(dict(name="i64_to_f64_to_utf8_chain", source="arrow-cast/src/cast/mod.rs:8449-8480 values through "
                       "display.rs:711-723 (ryu pretty layout; digits pinned by the Float64 expectations above)",
                       values=arr("Float64", [-9223372036854775808.0, -2147483648.0, 0.0, 127.0, 9223372036854775808.0]),
                       to="Utf8",
                       expected=arr("Utf8", ["-9.223372036854776e18", "-2147483648.0", "0.0", "127.0",
                                             "9.223372036854776e18"])))

# The reference's numeric cast matrices (arrow-cast/src/cast/mod.rs:7821-8990): one array of range-edge values per
# source type, cast to every numeric type in safe mode; out-of-range results are null.  `get_cast_values` (:8965)
# prints each value with Rust `{:?}`, i.e. shortest round-trip digits, so a printed float pins exactly one value
# ("-2147483600.0" as f32 IS -2147483648f32): the strings are transcribed as literals and parsed at the target width.
U64MAX = 18446744073709551615
CAST_MATRIX = [
    ("test_cast_from_f64", "7821-7989", "Float64",
     [-9223372036854775808.0, -2147483648.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967295.0, 18446744073709551616.0], {
        "Float64": [-9223372036854776000.0, -2147483648.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967295.0,
                    18446744073709552000.0],
        "Float32": [-9223372000000000000.0, -2147483600.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967300.0,
                    18446744000000000000.0],
        "Int64": [-9223372036854775808, -2147483648, -32768, -128, 0, 255, 65535, 4294967295, N],
        "Int32": [N, -2147483648, -32768, -128, 0, 255, 65535, N, N],
        "Int16": [N, N, -32768, -128, 0, 255, N, N, N],
        "Int8": [N, N, N, -128, 0, N, N, N, N],
        "UInt64": [N, N, N, N, 0, 255, 65535, 4294967295, N],
        "UInt32": [N, N, N, N, 0, 255, 65535, 4294967295, N],
        "UInt16": [N, N, N, N, 0, 255, 65535, N, N],
        "UInt8": [N, N, N, N, 0, 255, N, N, N]}),
    ("test_cast_from_f32", "7991-8134", "Float32",
     [-2147483648.0, -2147483648.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967296.0, 4294967296.0], {
        "Float64": [-2147483648.0, -2147483648.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967296.0, 4294967296.0],
        "Float32": [-2147483600.0, -2147483600.0, -32768.0, -128.0, 0.0, 255.0, 65535.0, 4294967300.0, 4294967300.0],
        "Int64": [-2147483648, -2147483648, -32768, -128, 0, 255, 65535, 4294967296, 4294967296],
        "Int32": [-2147483648, -2147483648, -32768, -128, 0, 255, 65535, N, N],
        "Int16": [N, N, -32768, -128, 0, 255, N, N, N],
        "Int8": [N, N, N, -128, 0, N, N, N, N],
        "UInt64": [N, N, N, N, 0, 255, 65535, 4294967296, 4294967296],
        "UInt32": [N, N, N, N, 0, 255, 65535, N, N],
        "UInt16": [N, N, N, N, 0, 255, 65535, N, N],
        "UInt8": [N, N, N, N, 0, 255, N, N, N]}),
    ("test_cast_from_uint64", "8136-8228", "UInt64", [0, 255, 65535, 4294967295, U64MAX], {
        "Float64": [0.0, 255.0, 65535.0, 4294967295.0, 18446744073709552000.0],
        "Float32": [0.0, 255.0, 65535.0, 4294967300.0, 18446744000000000000.0],
        "Int64": [0, 255, 65535, 4294967295, N],
        "Int32": [0, 255, 65535, N, N],
        "Int16": [0, 255, N, N, N],
        "Int8": [0, N, N, N, N],
        "UInt64": [0, 255, 65535, 4294967295, U64MAX],
        "UInt32": [0, 255, 65535, 4294967295, N],
        "UInt16": [0, 255, 65535, N, N],
        "UInt8": [0, 255, N, N, N]}),
    ("test_cast_from_uint32", "8230-8312", "UInt32", [0, 255, 65535, 4294967295], {
        "Float64": [0.0, 255.0, 65535.0, 4294967295.0],
        "Float32": [0.0, 255.0, 65535.0, 4294967300.0],
        "Int64": [0, 255, 65535, 4294967295],
        "Int32": [0, 255, 65535, N],
        "Int16": [0, 255, N, N],
        "Int8": [0, N, N, N],
        "UInt64": [0, 255, 65535, 4294967295],
        "UInt32": [0, 255, 65535, 4294967295],
        "UInt16": [0, 255, 65535, N],
        "UInt8": [0, 255, N, N]}),
    ("test_cast_from_uint16", "8314-8380", "UInt16", [0, 255, 65535], {
        "Float64": [0.0, 255.0, 65535.0], "Float32": [0.0, 255.0, 65535.0],
        "Int64": [0, 255, 65535], "Int32": [0, 255, 65535], "Int16": [0, 255, N], "Int8": [0, N, N],
        "UInt64": [0, 255, 65535], "UInt32": [0, 255, 65535], "UInt16": [0, 255, 65535], "UInt8": [0, 255, N]}),
    ("test_cast_from_uint8", "8382-8447", "UInt8", [0, 255], {
        "Float64": [0.0, 255.0], "Float32": [0.0, 255.0],
        "Int64": [0, 255], "Int32": [0, 255], "Int16": [0, 255], "Int8": [0, N],
        "UInt64": [0, 255], "UInt32": [0, 255], "UInt16": [0, 255], "UInt8": [0, 255]}),
    ("test_cast_from_int64", "8449-8590", "Int64",
     [I64MIN, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, I64MAX], {
        "Float32": [-9223372000000000000.0, -2147483600.0, -32768.0, -128.0, 0.0, 127.0, 32767.0, 2147483600.0,
                    9223372000000000000.0],
        "Int64": [I64MIN, -2147483648, -32768, -128, 0, 127, 32767, 2147483647, I64MAX],
        "Int16": [N, N, -32768, -128, 0, 127, 32767, N, N],
        "Int8": [N, N, N, -128, 0, 127, N, N, N],
        "UInt64": [N, N, N, N, 0, 127, 32767, 2147483647, I64MAX],
        "UInt32": [N, N, N, N, 0, 127, 32767, 2147483647, N],
        "UInt16": [N, N, N, N, 0, 127, 32767, N, N]}),
    ("test_cast_from_int32", "8592-8710", "Int32", [-2147483648, -32768, -128, 0, 127, 32767, 2147483647], {
        "Float64": [-2147483648.0, -32768.0, -128.0, 0.0, 127.0, 32767.0, 2147483647.0],
        "Float32": [-2147483600.0, -32768.0, -128.0, 0.0, 127.0, 32767.0, 2147483600.0],
        "Int16": [N, -32768, -128, 0, 127, 32767, N],
        "Int8": [N, N, -128, 0, 127, N, N],
        "UInt64": [N, N, N, 0, 127, 32767, 2147483647],
        "UInt32": [N, N, N, 0, 127, 32767, 2147483647],
        "UInt16": [N, N, N, 0, 127, 32767, N],
        "UInt8": [N, N, N, 0, 127, N, N]}),
    ("test_cast_from_int16", "8712-8803", "Int16", [-32768, -128, 0, 127, 32767], {
        "Float64": [-32768.0, -128.0, 0.0, 127.0, 32767.0], "Float32": [-32768.0, -128.0, 0.0, 127.0, 32767.0],
        "Int64": [-32768, -128, 0, 127, 32767], "Int32": [-32768, -128, 0, 127, 32767],
        "Int16": [-32768, -128, 0, 127, 32767], "Int8": [N, -128, 0, 127, N],
        "UInt64": [N, N, 0, 127, 32767], "UInt32": [N, N, 0, 127, 32767], "UInt16": [N, N, 0, 127, 32767],
        "UInt8": [N, N, 0, 127, N]}),
    ("test_cast_from_int8", "8830-8958", "Int8", [-128, 0, 127], {
        "Float64": [-128.0, 0.0, 127.0], "Float32": [-128.0, 0.0, 127.0],
        "Int64": [-128, 0, 127], "Int32": [-128, 0, 127], "Int16": [-128, 0, 127], "Int8": [-128, 0, 127],
        "UInt64": [N, 0, 127], "UInt32": [N, 0, 127], "UInt16": [N, 0, 127], "UInt8": [N, 0, 127]}),
]
for tname, lines, src_t, src, targets in CAST_MATRIX:
    for to_t, exp in targets.items():
        cast_cases.append(dict(name=f"{tname}_to_{to_t}", source=f"arrow-cast/src/cast/mod.rs:{lines}",
                               values=arr(src_t, src), to=to_t, expected=arr(to_t, exp)))
# test_cast_int32_to_u8_with_error (:4715): the unsafe form reports the first value that does not fit
cast_cases.append(dict(name="test_cast_int32_to_u8_with_error", source="arrow-cast/src/cast/mod.rs:4713-4726 (#[should_panic(expected = ...)])",
                       values=arr("Int32", [-5, 6, -7, 8, 100000000]), to="UInt8", safe=False, error="CastError",
                       message="Can't cast value -5 to type UInt8"))
cast_cases.append(dict(name="test_cast_i32_to_u8_sliced", source="arrow-cast/src/cast/mod.rs:4728-4739",
                       values=arr("Int32", [-5, 6, -7, 8, 100000000], [2, 3]), to="UInt8",
                       expected=arr("UInt8", [N, 8, N])))
cast_cases.append(dict(name="test_cast_i32_to_f64", source="arrow-cast/src/cast/mod.rs:4688-4698",
                       values=arr("Int32", [5, 6, 7, 8, 9]), to="Float64", expected=arr("Float64", [5.0, 6.0, 7.0, 8.0, 9.0])))
for st in ("Utf8", "LargeUtf8"):
    cast_cases.append(dict(name=f"test_cast_to_strings_{st}", source="arrow-cast/src/cast/mod.rs:7450-7470",
                           values=arr("Int32", [1, 2, 3]), to=st, expected=arr(st, ["1", "2", "3"])))
cast_cases.append(dict(name="test_cast_i32_to_i32", source="arrow-cast/src/cast/mod.rs:4742-4751",
                       values=arr("Int32", [5, 6, 7, 8, 9]), to="Int32", expected=arr("Int32", [5, 6, 7, 8, 9])))
# test_cast_bool_to_i32 / _to_f64 (:5006, :5046) and test_cast_i32_to_bool-style arms (cast/mod.rs:254-255)
cast_cases.append(dict(name="test_cast_bool_to_i32", source="arrow-cast/src/cast/mod.rs:5006-5014",
                       values=arr("Boolean", [T, F, N]), to="Int32", expected=arr("Int32", [1, 0, N])))
cast_cases.append(dict(name="test_cast_bool_to_f64", source="arrow-cast/src/cast/mod.rs:5046-5054",
                       values=arr("Boolean", [T, F, N]), to="Float64", expected=arr("Float64", [1.0, 0.0, N])))

# Utf8 / LargeUtf8 -> numeric: Parser::parse through parse_string (arrow-cast/src/cast/string.rs:66-120)
for st in ("Utf8", "LargeUtf8"):
    cast_cases.append(dict(name=f"test_cast_utf8_to_i32_{st}", source="arrow-cast/src/cast/mod.rs:4879-4889",
                           values=arr(st, ["5", "6", "seven", "8", "9.1"]), to="Int32", expected=arr("Int32", [5, 6, N, 8, N])))
    # test_cast_utf8view_to_f32 / test_cast_string_to_f16 (:4904, :4915) use these texts; 4.56 and 8.9 are the
    # nearest Float32 values, written here as their exact decimal expansions
    cast_cases.append(dict(name=f"test_cast_string_to_f32_{st}", source="arrow-cast/src/cast/mod.rs:4904-4913",
                           values=arr(st, ["3", "4.56", "seven", "8.9"]), to="Float32",
                           expected=arr("Float32", [3.0, 4.559999942779541015625, N, 8.89999961853027343750])))
    cast_cases.append(dict(name=f"test_cast_string_to_integral_overflow_{st}", source="arrow-cast/src/cast/mod.rs:5337-5354",
                           values=arr(st, ["123", "-123", "86374", N]), to="Int16", expected=arr("Int16", [123, -123, N, N])))
cast_cases.append(dict(name="test_cast_with_options_utf8_to_i32", source="arrow-cast/src/cast/mod.rs:4944-4964",
                       values=arr("Utf8", ["5", "6", "seven", "8", "9.1"]), to="Int32", safe=False, error="CastError",
                       message="Cannot cast string 'seven' to value of Int32 type"))
# test_parse_empty (parse.rs:2882-2897) and test_parse_prefix_white_space (:2911-2955), one array per target type
cast_cases.append(dict(name="test_parse_empty_ints", source="arrow-cast/src/parse.rs:2882-2897",
                       values=arr("Utf8", ["", "+"]), to="Int32", expected=arr("Int32", [N, N])))
for to_t in ("Int64", "UInt32", "UInt64"):
    cast_cases.append(dict(name=f"test_parse_empty_{to_t}", source="arrow-cast/src/parse.rs:2882-2897",
                           values=arr("Utf8", ["", "+"]), to=to_t, expected=arr(to_t, [N, N])))
for to_t in ("Float32", "Float64"):
    cast_cases.append(dict(name=f"test_parse_empty_{to_t}", source="arrow-cast/src/parse.rs:2882-2897",
                           values=arr("Utf8", ["", "+"]), to=to_t, expected=arr(to_t, [N, N])))
cast_cases.append(dict(name="test_parse_white_space_f64", source="arrow-cast/src/parse.rs:2911-2955",
                       values=arr("Utf8", [" 1.5", "\t\n 20.54", "\n2.5", "\n-942.5423", "\n\t\n\t\n40.5123", " 1.5",
                                           "\n\t\n\t\n-40.5123", " -1.5", "1.5 ", "40.5123\n", "40.5123\n\t\n\t\n", "-942.5423\t",
                                           " 1.5 ", "\t\n 20.54 \t", "\n-942.5423\n", "1.5abc", "40.5123x"]), to="Float64",
                       expected=arr("Float64", [1.5, 20.54, 2.5, -942.5423, 40.5123, 1.5, -40.5123, -1.5, 1.5, 40.5123, 40.5123,
                                                -942.5423, 1.5, 20.54, -942.5423, N, N])))
cast_cases.append(dict(name="test_parse_white_space_i32", source="arrow-cast/src/parse.rs:2911-2955",
                       values=arr("Utf8", [" 3", "          30", "\n \n 100", " \n25", "\t800", "\t  \n \t 851", "\t\n\t\n\n\n\t1",
                                           " \n-25", "\t-800", "3 ", "30          ", "-25 \n", "800\t", " 3 ", "\n \n 100 \n",
                                           "\t-800\t\n", "30x", "100px", "-25!", "3j", "3"]), to="Int32",
                       expected=arr("Int32", [3, 30, 100, 25, 800, 851, 1, -25, -800, 3, 30, -25, 800, 3, 100, -800, N, N, N, N, 3])))

bool_cases = []
A4, B4 = [F, F, T, T], [F, T, F, T]
bool_cases.append(dict(name="test_bool_array_and", source="arrow-arith/src/boolean.rs:364", op="and",
                       lhs=arr("Boolean", A4), rhs=arr("Boolean", B4), expected=arr("Boolean", [F, F, F, T])))
bool_cases.append(dict(name="test_bool_array_or", source="arrow-arith/src/boolean.rs:375", op="or",
                       lhs=arr("Boolean", A4), rhs=arr("Boolean", B4), expected=arr("Boolean", [F, T, T, T])))
bool_cases.append(dict(name="test_bool_array_and_not", source="arrow-arith/src/boolean.rs:386", op="and_not",
                       lhs=arr("Boolean", A4), rhs=arr("Boolean", B4), expected=arr("Boolean", [F, F, T, F])))
A9 = [N, N, N, F, F, F, T, T, T]
B9 = [N, F, T, N, F, T, N, F, T]
bool_cases.append(dict(name="test_bool_array_or_nulls", source="arrow-arith/src/boolean.rs:422", op="or",
                       lhs=arr("Boolean", A9), rhs=arr("Boolean", B9), expected=arr("Boolean", [N, N, N, N, F, T, N, T, T])))
bool_cases.append(dict(name="test_bool_array_and_kleene_nulls", source="arrow-arith/src/boolean.rs:473", op="and_kleene",
                       lhs=arr("Boolean", A9), rhs=arr("Boolean", B9), expected=arr("Boolean", [N, F, N, F, F, F, N, F, T])))
bool_cases.append(dict(name="test_bool_array_or_kleene_nulls", source="arrow-arith/src/boolean.rs:514", op="or_kleene",
                       lhs=arr("Boolean", A9), rhs=arr("Boolean", B9), expected=arr("Boolean", [N, N, T, N, F, T, T, T, T])))
bool_cases.append(dict(name="and_kleene_doc_example", source="arrow-arith/src/boolean.rs:47-51", op="and_kleene",
                       lhs=arr("Boolean", [T, F, N]), rhs=arr("Boolean", [N, N, N]), expected=arr("Boolean", [N, F, N])))
bool_cases.append(dict(name="test_boolean_array_kleene_no_remainder", source="arrow-arith/src/boolean.rs:451-458", op="or_kleene",
                       lhs=arr("Boolean", [T] * 1024), rhs=arr("Boolean", [N] * 1024), expected=arr("Boolean", [T] * 1024)))
bool_cases.append(dict(name="test_bool_array_not", source="arrow-arith/src/boolean.rs:621", op="not",
                       lhs=arr("Boolean", [F, T]), expected=arr("Boolean", [T, F])))
bool_cases.append(dict(name="test_bool_array_not_sliced", source="arrow-arith/src/boolean.rs:631", op="not",
                       lhs=arr("Boolean", [N, T, F, N, T], [1, 4]), expected=arr("Boolean", [F, T, N, F])))
bool_cases.append(dict(name="test_nullable_array_is_null", source="arrow-arith/src/boolean.rs:835", op="is_null",
                       lhs=arr("Int32", [1, N, 3, N]), expected=arr("Boolean", [F, T, F, T]), no_null_buffer=True))
bool_cases.append(dict(name="test_nullable_array_is_not_null", source="arrow-arith/src/boolean.rs:878", op="is_not_null",
                       lhs=arr("Int32", [1, N, 3, N]), expected=arr("Boolean", [T, F, T, F]), no_null_buffer=True))
bool_cases.append(dict(name="length_mismatch", source="arrow-arith/src/boolean.rs:233-237", op="and",
                       lhs=arr("Boolean", [T, F]), rhs=arr("Boolean", [T]), error="ComputeError",
                       message="Cannot perform bitwise operation on arrays of different length"))

bool_cases.append(dict(name="test_nullif_int_array", source="arrow-select/src/nullif.rs:126-142", op="nullif",
                       lhs=arr("Int32", [15, N, 8, 1, 9]), rhs=arr("Boolean", [F, N, T, F, N]),
                       expected=arr("Int32", [15, N, N, 1, 9])))
bool_cases.append(dict(name="test_nullif_int_array_offset", source="arrow-select/src/nullif.rs:160-190", op="nullif",
                       lhs=arr("Int32", [N, 15, 8, 1, 9], [1, 3]), rhs=arr("Boolean", [F, F, F, N, T, F, N], [2, 3]),
                       expected=arr("Int32", [15, 8, N])))
bool_cases.append(dict(name="test_nullif_string", source="arrow-select/src/nullif.rs:192-215", op="nullif",
                       lhs=arr("Utf8", ["hello", N, "world", "a", "b"]), rhs=arr("Boolean", [T, T, F, T, N]),
                       expected=arr("Utf8", [N, N, "world", N, "b"])))

# round 6: the rest of the in-scope inline vectors of boolean.rs / nullif.rs (slices, bit offsets, one-sided null buffers)
BL = "arrow-arith/src/boolean.rs"
# :398 / :410 assert and_not(a, b) == and(a, not(b)) on slices: expected = a AND NOT b, row by row
bool_cases.append(dict(name="test_bool_array_and_not_sliced", source=f"{BL}:398-407", op="and_not",
                       lhs=arr("Boolean", [T, F, T, F, T, F, T], [2, 3]), rhs=arr("Boolean", [F, T, F, T, F, T, F], [2, 3]),
                       expected=arr("Boolean", [T, F, T])))
bool_cases.append(dict(name="test_bool_array_and_not_sliced_different_offsets", source=f"{BL}:410-419", op="and_not",
                       lhs=arr("Boolean", [F, T, T, F, T, F, T], [1, 4]), rhs=arr("Boolean", [T, F, F, T, F, T, F], [2, 4]),
                       expected=arr("Boolean", [T, F, F, F])))
bool_cases.append(dict(name="test_bool_array_or_kleene_right_sided_nulls", source=f"{BL}:555-584", op="or_kleene",
                       lhs=arr("Boolean", [F, F, F, T, T, T]), rhs=arr("Boolean", [T, F, N, T, F, N]),
                       expected=arr("Boolean", [T, F, N, T, T, T])))
bool_cases.append(dict(name="test_bool_array_or_kleene_left_sided_nulls", source=f"{BL}:587-616", op="or_kleene",
                       lhs=arr("Boolean", [T, F, N, T, F, N]), rhs=arr("Boolean", [F, F, F, T, T, T]),
                       expected=arr("Boolean", [T, F, N, T, T, T])))
bool_cases.append(dict(name="test_bool_array_and_nulls", source=f"{BL}:642-678", op="and",
                       lhs=arr("Boolean", A9), rhs=arr("Boolean", B9), expected=arr("Boolean", [N, N, N, N, F, F, N, F, T])))
F8 = [F] * 8
bool_cases.append(dict(name="test_bool_array_and_sliced_same_offset", source=f"{BL}:680-698", op="and",
                       lhs=arr("Boolean", F8 + [F, F, T, T], [8, 4]), rhs=arr("Boolean", F8 + [F, T, F, T], [8, 4]),
                       expected=arr("Boolean", [F, F, F, T])))
bool_cases.append(dict(name="test_bool_array_and_sliced_same_offset_mod8", source=f"{BL}:700-718", op="and",
                       lhs=arr("Boolean", [F, F, T, T] + F8, [0, 4]), rhs=arr("Boolean", F8 + [F, T, F, T], [8, 4]),
                       expected=arr("Boolean", [F, F, F, T])))
bool_cases.append(dict(name="test_bool_array_and_sliced_offset1", source=f"{BL}:720-734", op="and",
                       lhs=arr("Boolean", F8 + [F, F, T, T], [8, 4]), rhs=arr("Boolean", [F, T, F, T]),
                       expected=arr("Boolean", [F, F, F, T])))
bool_cases.append(dict(name="test_bool_array_and_sliced_offset2", source=f"{BL}:736-750", op="and",
                       lhs=arr("Boolean", [F, F, T, T]), rhs=arr("Boolean", F8 + [F, T, F, T], [8, 4]),
                       expected=arr("Boolean", [F, F, F, T])))
bool_cases.append(dict(name="test_bool_array_and_nulls_offset", source=f"{BL}:752-773", op="and",
                       lhs=arr("Boolean", [N, F, T, N, T], [1, 4]), rhs=arr("Boolean", [N, N, T, F, T, T], [2, 4]),
                       expected=arr("Boolean", [F, F, N, T])))
I15 = [1, 2, 3, 4, 5, 6, 7, 8, 7, 6, 5, 4, 3, 2, 1]
bool_cases.append(dict(name="test_nonnull_array_is_null", source=f"{BL}:775-785", op="is_null",
                       lhs=arr("Int32", [1, 2, 3, 4]), expected=arr("Boolean", [F, F, F, F]), no_null_buffer=True))
bool_cases.append(dict(name="test_nonnull_array_with_offset_is_null", source=f"{BL}:787-798", op="is_null",
                       lhs=arr("Int32", I15, [8, 4]), expected=arr("Boolean", [F, F, F, F]), no_null_buffer=True))
bool_cases.append(dict(name="test_nonnull_array_is_not_null", source=f"{BL}:800-810", op="is_not_null",
                       lhs=arr("Int32", [1, 2, 3, 4]), expected=arr("Boolean", [T, T, T, T]), no_null_buffer=True))
bool_cases.append(dict(name="test_nonnull_array_with_offset_is_not_null", source=f"{BL}:812-823", op="is_not_null",
                       lhs=arr("Int32", I15, [8, 4]), expected=arr("Boolean", [T, T, T, T]), no_null_buffer=True))
N8I = [N] * 8 + [1, N, 2, N, 3, 4, N, N]
bool_cases.append(dict(name="test_nullable_array_with_offset_is_null", source=f"{BL}:846-874", op="is_null",
                       lhs=arr("Int32", N8I, [8, 4]), expected=arr("Boolean", [F, T, F, T]), no_null_buffer=True))
bool_cases.append(dict(name="test_nullable_array_with_offset_is_not_null", source=f"{BL}:890-918", op="is_not_null",
                       lhs=arr("Int32", N8I, [8, 4]), expected=arr("Boolean", [T, F, T, F]), no_null_buffer=True))
NF = "arrow-select/src/nullif.rs"
bool_cases.append(dict(name="test_nullif_int_large_left_offset", source=f"{NF}:229-275", op="nullif",
                       lhs=arr("Int32", [-1] * 16 + [N, 15, 8, 1, 9], [17, 3]), rhs=arr("Boolean", [F, F, F, N, T, F, N], [2, 3]),
                       expected=arr("Int32", [15, 8, N])))
bool_cases.append(dict(name="test_nullif_int_large_right_offset", source=f"{NF}:278-324", op="nullif",
                       lhs=arr("Int32", [N, 15, 8, 1, 9], [1, 3]), rhs=arr("Boolean", [F] * 19 + [N, T, F, N], [18, 3]),
                       expected=arr("Int32", [15, 8, N])))
bool_cases.append(dict(name="test_nullif_boolean_offset", source=f"{NF}:327-358", op="nullif",
                       lhs=arr("Boolean", [N, T, F, T, T], [1, 3]), rhs=arr("Boolean", [F, F, F, N, T, F, N], [2, 3]),
                       expected=arr("Boolean", [T, F, N])))
bool_cases.append(dict(name="test_nullif_no_nulls", source=f"{NF}:464-472", op="nullif",
                       lhs=arr("Int32", [15, 7, 8, 1, 9]), rhs=arr("Boolean", [F, N, T, F, N]),
                       expected=arr("Int32", [15, 7, N, 1, 9])))
bool_cases.append(dict(name="nullif_empty", source=f"{NF}:475-480", op="nullif",
                       lhs=arr("Int32", []), rhs=arr("Boolean", []), expected=arr("Int32", [])))


# ---------------------------------------------------------------- aggregate (arrow-arith/src/aggregate.rs tests)
agg_cases = []
A_ = "arrow-arith/src/aggregate.rs"


def agg(name, line, op, values, expected, **kw):
    agg_cases.append(dict(name=name, source=f"{A_}:{line}", op=op, values=values, expected=expected, **kw))


agg("test_primitive_array_sum", 1039, "sum", arr("Int32", [1, 2, 3, 4, 5]), 15)
agg("test_primitive_array_float_sum", 1045, "sum", arr("Float64", [1.1, 2.2, 3.3, 4.4, 5.5]), 16.5)
agg("test_primitive_array_product", 1051, "product", arr("Int32", [1, 2, 3, 4, 5]), 120)
agg("test_primitive_array_float_product", 1057, "product", arr("Float64", [1.0, 2.0, 3.0, 4.0, 5.0]), 120.0)
agg("test_primitive_array_product_with_nulls", 1063, "product", arr("Int32", [N, 2, 3, N, 5]), 30)
agg("test_primitive_array_product_all_nulls", 1069, "product", arr("Int32", [N, N, N]), None)
agg("test_primitive_array_product_empty", 1075, "product", arr("Int32", []), None)
agg("test_primitive_array_product_checked", 1081, "product_checked", arr("Int32", [1, 2, 3, 4, 5]), 120)
agg("test_primitive_array_product_checked_with_nulls", 1087, "product_checked", arr("Int32", [N, 2, 3, N, 5]), 30)
agg("test_primitive_array_product_checked_all_nulls", 1093, "product_checked", arr("Int32", [N, N, N]), None)
agg("test_product_overflow", 1099, "product", arr("Int32", [2147483647, 2]), -2)
agg("test_product_checked_overflow", 1106, "product_checked", arr("Int32", [2147483647, 2]), None,
    error="ArithmeticOverflow", message="Overflow happened on: 2147483647 * 2")
agg("test_primitive_array_sum_with_nulls", 1112, "sum", arr("Int32", [N, 2, 3, N, 5]), 10)
agg("test_primitive_array_sum_all_nulls", 1118, "sum", arr("Int32", [N, N, N]), None)
third = [bool(x % 3 == 0) for x in range(1, 101)]
for t, line in (("Float64", 1124), ("Float32", 1139), ("Int64", 1154), ("Int32", 1166), ("Int16", 1177)):
    fl = t.startswith("Float")
    vals = [float(x) if fl else x for x in range(1, 101)]
    agg(f"test_primitive_array_sum_large_{t}", line, "sum", arr(t, vals), float(5050) if fl else 5050)
    # "non-zero values at the invalid indices": raw values + validity
    tot = sum(x for x in range(1, 101) if x % 3 == 0)
    agg(f"test_primitive_array_sum_large_{t}_nullable", line, "sum", dict(type=t, raw=vals, valid=third),
        float(tot) if fl else tot)
agg("test_primitive_array_sum_large_8", 1188, "sum", arr("UInt8", list(range(1, 101))), 5050 % 256)
agg("test_primitive_array_sum_large_8_nullable", 1188, "sum", dict(type="UInt8", raw=list(range(1, 101)), valid=third),
    sum(x for x in range(1, 101) if x % 3 == 0) % 256)
agg("test_primitive_array_bit_and", 1211, "bit_and", arr("Int32", [1, 2, 3, 4, 5]), 0)
agg("test_primitive_array_bit_and_with_nulls", 1217, "bit_and", arr("Int32", [N, 2, 3, N, N]), 2)
agg("test_primitive_array_bit_and_all_nulls", 1223, "bit_and", arr("Int32", [N, N, N]), None)
agg("test_primitive_array_bit_or", 1229, "bit_or", arr("Int32", [1, 2, 3, 4, 5]), 7)
agg("test_primitive_array_bit_or_with_nulls", 1235, "bit_or", arr("Int32", [N, 2, 3, N, 5]), 7)
agg("test_primitive_array_bit_or_all_nulls", 1241, "bit_or", arr("Int32", [N, N, N]), None)
agg("test_primitive_array_bit_xor", 1247, "bit_xor", arr("Int32", [1, 2, 3, 4, 5]), 1)
agg("test_primitive_array_bit_xor_with_nulls", 1253, "bit_xor", arr("Int32", [N, 2, 3, N, 5]), 4)
agg("test_primitive_array_bit_xor_all_nulls", 1259, "bit_xor", arr("Int32", [N, N, N]), None)
agg("test_primitive_array_bool_and", 1265, "min", arr("Boolean", [T, F, T, F, T]), False)
agg("test_primitive_array_bool_and_with_nulls", 1271, "min", arr("Boolean", [N, T, T, N, T]), True)
agg("test_primitive_array_bool_and_all_nulls", 1277, "min", arr("Boolean", [N, N, N]), None)
agg("test_primitive_array_bool_or", 1283, "max", arr("Boolean", [T, F, T, F, T]), True)
agg("test_primitive_array_bool_or_with_nulls", 1289, "max", arr("Boolean", [N, F, F, N, F]), False)
agg("test_primitive_array_bool_or_all_nulls", 1295, "max", arr("Boolean", [N, N, N]), None)
for op, want in (("min", 5), ("max", 9)):
    agg(f"test_primitive_array_min_max_{op}", 1301, op, arr("Int32", [5, 6, 7, 8, 9]), want)
    agg(f"test_primitive_array_min_max_with_nulls_{op}", 1308, op, arr("Int32", [5, N, N, 8, 9]), want)
agg("test_primitive_min_max_1_min", 1315, "min", arr("Int32", [N, N, 5, 2]), 2)
agg("test_primitive_min_max_1_max", 1315, "max", arr("Int32", [N, N, 5, 2]), 5)
f256 = [float(i + 1) for i in range(256)]
agg("float_large_nonnull_min", 1322, "min", arr("Float64", f256), 1.0)
agg("float_large_nonnull_max", 1322, "max", arr("Float64", f256), 256.0)
agg("float_large_nonnull_max_255", 1322, "max", arr("Float64", f256[:255]), 255.0)
agg("float_large_nonnull_max_257", 1322, "max", arr("Float64", f256 + [257.0]), 257.0)
n3 = [None if (i + 1) % 3 == 0 else float(i + 1) for i in range(256)]
agg("float_large_nullable_min", 1336, "min", arr("Float64", n3), 1.0)
agg("float_large_nullable_max", 1336, "max", arr("Float64", n3), 256.0)
edge = [None if i in (0, 255) else float(i + 1) for i in range(256)]
agg("float_large_nullable_boundary_nulls_min", 1350, "min", arr("Float64", edge), 2.0)
agg("float_large_nullable_boundary_nulls_max", 1350, "max", arr("Float64", edge), 255.0)
single = [float(i) if i == 100 else None for i in range(256)]
agg("float_large_nullable_single_min", 1363, "min", arr("Float64", single), 100.0)
agg("float_large_nullable_single_max", 1363, "max", arr("Float64", single), 100.0)
for nm, v in (("neg_inf", "-inf"), ("f64_min", -1.7976931348623157e308), ("f64_max", 1.7976931348623157e308), ("inf", "inf")):
    for op in ("min", "max"):
        agg(f"float_edge_{nm}_{op}", 1381, op, arr("Float64", [v] * 100), v)
for op in ("min", "max"):
    agg(f"float_all_nans_{op}", 1400, op, arr("Float64", ["nan"] * 100), "nan")
# test_primitive_min_max_float_negative_nan (:1407): max is +NaN, min is -NaN (total order)
agg("float_negative_nan_max", 1407, "max", arr("Float64", ["-inf", "nan", "inf", "-nan"]), "nan")
agg("float_negative_nan_min", 1407, "min", arr("Float64", ["-inf", "nan", "inf", "-nan"]), "-nan")
first = ["nan"] + [float(i) for i in range(1, 100)]
last = [float(i + 1) for i in range(99)] + ["nan"]
agg("float_first_nan_nonnull_min", 1420, "min", arr("Float64", first), 1.0)
agg("float_first_nan_nonnull_max", 1420, "max", arr("Float64", first), "nan")
agg("float_last_nan_nonnull_min", 1435, "min", arr("Float64", last), 1.0)
agg("float_last_nan_nonnull_max", 1435, "max", arr("Float64", last), "nan")
firstn = ["nan" if i == 0 else (None if i % 2 == 0 else float(i)) for i in range(100)]
lastn = ["nan" if i == 99 else (None if i % 2 == 0 else float(i)) for i in range(100)]
agg("float_first_nan_nullable_min", 1450, "min", arr("Float64", firstn), 1.0)
agg("float_first_nan_nullable_max", 1450, "max", arr("Float64", firstn), "nan")
agg("float_last_nan_nullable_min", 1467, "min", arr("Float64", lastn), 1.0)
agg("float_last_nan_nullable_max", 1467, "max", arr("Float64", lastn), "nan")
mix = [{0: "-inf", 1: -1.7976931348623157e308, 2: 1.7976931348623157e308, 4: "inf", 5: "nan"}.get(i % 10, float(i))
       for i in range(100)]
agg("float_inf_and_nans_min", 1484, "min", arr("Float64", mix), "-inf")
agg("float_inf_and_nans_max", 1484, "max", arr("Float64", mix), "nan")


# round 6: min_boolean / max_boolean (:1670-1825), sliced inputs (:1896-1938), wrapping / checked sum at the overflow (:1984-1998)
def agg_minmax(name, line, values, lo, hi):
    agg(f"{name}_min", line, "min", values, lo)
    agg(f"{name}_max", line, "max", values, hi)


agg_minmax("test_boolean_min_max_empty", 1670, arr("Boolean", []), None, None)
agg_minmax("test_boolean_min_max_all_null", 1677, arr("Boolean", [N, N]), None, None)
agg_minmax("test_boolean_min_max_no_null", 1684, arr("Boolean", [T, F, T]), False, True)
for k, (v, lo, hi) in enumerate([([T, T, N, F, N], False, True), ([N, T, N, F, N], False, True), ([F, T, N, F, N], False, True),
                                 ([T, N], True, True), ([F, N], False, False), ([T], True, True), ([F], False, False)]):
    agg_minmax(f"test_boolean_min_max_{k}", "1691-1719", arr("Boolean", v), lo, hi)
for k, (v, lo, hi) in enumerate([([F], False, False), ([N, F], False, False), ([N, T], True, True), ([T], True, True)]):
    agg_minmax(f"test_boolean_min_max_smaller_{k}", "1722-1738", arr("Boolean", v), lo, hi)
agg_minmax("test_boolean_min_max_64_true_64_false", 1741, arr("Boolean", [T] * 64 + [F] * 64), False, True)
agg_minmax("test_boolean_min_max_64_true_64_false_nulls", 1741, arr("Boolean", [T] * 31 + [N] + [T] * 32 + [F] + [N] * 63), False, True)
agg_minmax("test_boolean_min_max_64_false_64_true", 1763, arr("Boolean", [F] * 64 + [T] * 64), False, True)
agg_minmax("test_boolean_min_max_64_false_64_true_nulls", 1763, arr("Boolean", [F] * 31 + [N] + [F] * 32 + [T] + [N] * 63), False, True)
agg_minmax("test_boolean_min_max_96_true", 1785, arr("Boolean", [T] * 96), True, True)
agg_minmax("test_boolean_min_max_96_true_nulls", 1785, arr("Boolean", [T] * 31 + [N] + [T] * 32 + [T] * 31 + [N]), True, True)
agg_minmax("test_boolean_min_max_96_false", 1806, arr("Boolean", [F] * 96), False, False)
agg_minmax("test_boolean_min_max_96_false_nulls", 1806, arr("Boolean", [F] * 31 + [N] + [F] * 32 + [F] * 31 + [N]), False, False)
agg_minmax("test_min_max_sliced_primitive", 1896, arr("Float64", [N, N, N, N, N, 4.0], [4, 2]), 4.0, 4.0)
agg_minmax("test_min_max_sliced_boolean", 1918, arr("Boolean", [N, N, N, N, N, T], [4, 2]), True, True)
agg("test_sum_overflow", 1984, "sum", arr("Int32", [2147483647, 1]), -2147483648)
agg("test_sum_checked_overflow", 1992, "sum_checked", arr("Int32", [2147483647, 1]), None,
    error="ArithmeticOverflow", message="Overflow happened on: 2147483647 + 1")  # (add_checked's text, arrow-array/src/arithmetic.rs:167-175)

# ---------------------------------------------------------------- concat (arrow-select/src/concat.rs tests)
concat_cases = [
    dict(name="test_concat_string_arrays", source="arrow-select/src/concat.rs:832-853",
         pieces=[arr("Utf8", ["hello", "world"]), arr("Utf8", ["2", "3", "4"]), arr("Utf8", ["foo", "bar", N, "baz"])],
         expected=arr("Utf8", ["hello", "world", "2", "3", "4", "foo", "bar", N, "baz"])),
    dict(name="test_concat_large_string_arrays", source="arrow-select/src/concat.rs:832-853 (LargeUtf8 instantiation)",
         pieces=[arr("LargeUtf8", ["hello", "world"]), arr("LargeUtf8", ["2", "3", "4"]), arr("LargeUtf8", ["foo", "bar", N, "baz"])],
         expected=arr("LargeUtf8", ["hello", "world", "2", "3", "4", "foo", "bar", N, "baz"])),
    dict(name="test_concat_primitive_arrays", source="arrow-select/src/concat.rs:880-904",
         pieces=[arr("Int64", [-1, -1, 2, N, N]), arr("Int64", [101, 102, 103, N]), arr("Int64", [256, 512, 1024])],
         expected=arr("Int64", [-1, -1, 2, N, N, 101, 102, 103, N, 256, 512, 1024])),
    dict(name="test_concat_primitive_array_slices", source="arrow-select/src/concat.rs:907-927",
         pieces=[arr("Int64", [-1, -1, 2, N, N], [1, 3]), arr("Int64", [101, 102, 103, N], [1, 3])],
         expected=arr("Int64", [-1, 2, N, 102, 103, N])),
    dict(name="test_concat_boolean_primitive_arrays", source="arrow-select/src/concat.rs:930-958",
         pieces=[arr("Boolean", [T, T, F, N, N, F]), arr("Boolean", [N, F, T, F])],
         expected=arr("Boolean", [T, T, F, N, N, F, N, F, T, F])),
    dict(name="test_concat_incompatible_datatypes", source="arrow-select/src/concat.rs:731-746",
         pieces=[arr("Int64", [-1, 2, N]), arr("Utf8", ["hello", "bar", "world"]), arr("Utf8", ["hey", "", "you"]),
                 arr("Int32", [-1, 2, N])], error="InvalidArgumentError",
         message="It is not possible to concatenate arrays of different data types (Int64, Utf8, Int32)."),
    dict(name="test_concat_10_incompatible_datatypes_should_include_all_of_them", source="arrow-select/src/concat.rs:749-772",
         pieces=[arr("Int64", [-1, 2, N]), arr("Utf8", ["hello", "bar", "world"]), arr("Utf8", ["hey", "", "you"]),
                 arr("Int32", [-1, 2, N]), arr("Int8", [-1, 2, N]), arr("Int16", [-1, 2, N]), arr("UInt8", [1, 2, N]),
                 arr("UInt16", [1, 2, N]), arr("UInt32", [1, 2, N]), arr("UInt16", [1, 2, N]), arr("UInt64", [1, 2, N]),
                 arr("Float32", [1.0, 2.0, N])], error="InvalidArgumentError",
         message="It is not possible to concatenate arrays of different data types (Int64, Utf8, Int32, Int8, Int16, UInt8, "
                 "UInt16, UInt32, UInt64, Float32)."),
    dict(name="test_concat_11_incompatible_datatypes_should_only_include_10", source="arrow-select/src/concat.rs:775-799",
         pieces=[arr("Int64", [-1, 2, N]), arr("Utf8", ["hello", "bar", "world"]), arr("Utf8", ["hey", "", "you"]),
                 arr("Int32", [-1, 2, N]), arr("Int8", [-1, 2, N]), arr("Int16", [-1, 2, N]), arr("UInt8", [1, 2, N]),
                 arr("UInt16", [1, 2, N]), arr("UInt32", [1, 2, N]), arr("UInt16", [1, 2, N]), arr("UInt64", [1, 2, N]),
                 arr("Float32", [1.0, 2.0, N]), arr("Float64", [1.0, 2.0, N])], error="InvalidArgumentError",
         message="It is not possible to concatenate arrays of different data types (Int64, Utf8, Int32, Int8, Int16, UInt8, "
                 "UInt16, UInt32, UInt64, Float32, ...)."),
    dict(name="test_concat_13_incompatible_datatypes_should_not_include_all_of_them",
         source="arrow-select/src/concat.rs:803-829 (Float16 piece replaced by LargeUtf8: the message stops before it either way)",
         pieces=[arr("Int64", [-1, 2, N]), arr("Utf8", ["hello", "bar", "world"]), arr("Utf8", ["hey", "", "you"]),
                 arr("Int32", [-1, 2, N]), arr("Int8", [-1, 2, N]), arr("Int16", [-1, 2, N]), arr("UInt8", [1, 2, N]),
                 arr("UInt16", [1, 2, N]), arr("UInt32", [1, 2, N]), arr("UInt16", [1, 2, N]), arr("UInt64", [1, 2, N]),
                 arr("Float32", [1.0, 2.0, N]), arr("Float64", [1.0, 2.0, N]), arr("LargeUtf8", [N, N, N]),
                 arr("Boolean", [T, F, N])], error="InvalidArgumentError",
         message="It is not possible to concatenate arrays of different data types (Int64, Utf8, Int32, Int8, Int16, UInt8, "
                 "UInt16, UInt32, UInt64, Float32, ...)."),
    dict(name="string_slices", source="arrow-select/src/concat.rs:1140-1170 (test_string_array_slices recipe)",
         pieces=[arr("Utf8", ["hello", "A", "B", "C"], [1, 3]), arr("Utf8", ["D", "E", N, "F"], [2, 2])],
         expected=arr("Utf8", ["A", "B", "C", N, "F"])),
    dict(name="test_string_array_with_null_slices", source="arrow-select/src/concat.rs:1328-1340",
         pieces=[arr("Utf8", ["hello", N, "A", "C"], [1, 3]), arr("Utf8", [N, "world", "D", N], [1, 2])],
         expected=arr("Utf8", [N, "A", "C", "world", "D"])),
    dict(name="test_concat_one_element_vec", source="arrow-select/src/concat.rs:717-729",
         pieces=[arr("Int64", [-1, 2, N])], expected=arr("Int64", [-1, 2, N])),
]

# ---------------------------------------------------------------- sort_to_indices (arrow-ord/src/sort.rs tests)
sort_cases = []
S_ = "arrow-ord/src/sort.rs"
base_i = [N, 0, 2, -1, 0, N]
for t in ("Int8", "Int16", "Int32", "Int64"):
    sort_cases.append(dict(name=f"primitives_default_{t}", source=f"{S_}:1626-1649", values=arr(t, base_i),
                           expected=[0, 5, 3, 1, 4, 2]))
    sort_cases.append(dict(name=f"primitives_desc_{t}", source=f"{S_}:1690-1725", values=arr(t, base_i), descending=True,
                           nulls_first=False, expected=[2, 1, 4, 3, 0, 5]))
    sort_cases.append(dict(name=f"primitives_desc_nulls_first_{t}", source=f"{S_}:1769-1805", values=arr(t, base_i),
                           descending=True, nulls_first=True, expected=[0, 5, 2, 1, 4, 3]))
for t in ("Float32", "Float64"):
    sort_cases.append(dict(name=f"primitives_default_{t}", source=f"{S_}:1664-1687",
                           values=arr(t, [N, -0.05, 2.225, -1.01, -0.05, N]), expected=[0, 5, 3, 1, 4, 2]))
    sort_cases.append(dict(name=f"primitives_desc_{t}", source=f"{S_}:1741-1766",
                           values=arr(t, [N, 0.005, 20.22, -10.3, 0.005, N]), descending=True, nulls_first=False,
                           expected=[2, 1, 4, 3, 0, 5]))
    sort_cases.append(dict(name=f"primitives_desc_nulls_first_{t}", source=f"{S_}:1821-1850",
                           values=arr(t, [N, 0.1, 0.2, -1.3, 0.01, N]), descending=True, nulls_first=True,
                           expected=[0, 5, 2, 1, 4, 3]))
sort_cases += [
    dict(name="limit_valid_less_than_limit_nulls_last", source=f"{S_}:1853-1862", values=arr("Float64", [2.0, N, N, 1.0]),
         descending=False, nulls_first=False, limit=3, expected=[3, 0, 1]),
    dict(name="limit_valid_less_than_limit_nulls_first", source=f"{S_}:1864-1872", values=arr("Float64", [2.0, N, N, 1.0]),
         descending=False, nulls_first=True, limit=3, expected=[1, 2, 3]),
    dict(name="more_nulls_than_limit_nulls_first", source=f"{S_}:1875-1883", values=arr("Float64", [1.0, N, N, N]),
         descending=False, nulls_first=True, limit=2, expected=[1, 2]),
    dict(name="more_nulls_than_limit_nulls_last", source=f"{S_}:1885-1893", values=arr("Float64", [1.0, N, N, N]),
         descending=False, nulls_first=False, limit=2, expected=[0, 1]),
    dict(name="test_sort_to_indices_primitive_more_nulls_than_limit", source=f"{S_}:1897-1907",
         values=arr("Int32", [N, N, 3, N, 1, N, 2]), descending=False, nulls_first=False, limit=2, expected=[4, 6]),
]
bools = [N, F, T, T, F, N]
sort_cases += [
    dict(name="boolean_default", source=f"{S_}:1912-1917", values=arr("Boolean", bools), expected=[0, 5, 1, 4, 2, 3]),
    dict(name="boolean_desc", source=f"{S_}:1920-1928", values=arr("Boolean", bools), descending=True, nulls_first=False,
         expected=[2, 3, 1, 4, 0, 5]),
    dict(name="boolean_desc_nulls_first", source=f"{S_}:1931-1939", values=arr("Boolean", bools), descending=True,
         nulls_first=True, expected=[0, 5, 2, 3, 1, 4]),
    dict(name="boolean_desc_nulls_first_limit", source=f"{S_}:1942-1950", values=arr("Boolean", bools), descending=True,
         nulls_first=True, limit=3, expected=[0, 5, 2]),
    dict(name="boolean_limit_nulls_last", source=f"{S_}:1953-1961", values=arr("Boolean", [T, N, N, F]), descending=False,
         nulls_first=False, limit=3, expected=[3, 0, 1]),
    dict(name="boolean_limit_nulls_first", source=f"{S_}:1963-1971", values=arr("Boolean", [T, N, N, F]), descending=False,
         nulls_first=True, limit=3, expected=[1, 2, 3]),
    dict(name="boolean_more_nulls_than_limit", source=f"{S_}:1974-1982", values=arr("Boolean", [T, N, N, N]),
         descending=False, nulls_first=True, limit=2, expected=[1, 2]),
]

# test_sort_to_indices_strings (:3055-3180), for both offset widths
STR6 = [N, "bad", "sad", N, "glad", "-ad"]
for st in ("Utf8", "LargeUtf8"):
    sort_cases += [
        dict(name=f"strings_default_{st}", source=f"{S_}:3056-3068", values=arr(st, STR6), expected=[0, 3, 5, 1, 4, 2]),
        dict(name=f"strings_desc_nulls_last_{st}", source=f"{S_}:3070-3085", values=arr(st, STR6), descending=True, nulls_first=False,
             expected=[2, 4, 1, 5, 0, 3]),
        dict(name=f"strings_asc_nulls_first_{st}", source=f"{S_}:3087-3102", values=arr(st, STR6), descending=False, nulls_first=True,
             expected=[0, 3, 5, 1, 4, 2]),
        dict(name=f"strings_desc_nulls_first_{st}", source=f"{S_}:3104-3119", values=arr(st, STR6), descending=True, nulls_first=True,
             expected=[0, 3, 2, 4, 1, 5]),
        dict(name=f"strings_desc_nulls_first_limit_{st}", source=f"{S_}:3121-3136", values=arr(st, STR6), descending=True,
             nulls_first=True, limit=3, expected=[0, 3, 2]),
        dict(name=f"strings_limit_nulls_last_{st}", source=f"{S_}:3139-3148", values=arr(st, ["def", N, N, "abc"]),
             descending=False, nulls_first=False, limit=3, expected=[3, 0, 1]),
        dict(name=f"strings_limit_nulls_first_{st}", source=f"{S_}:3150-3158", values=arr(st, ["def", N, N, "abc"]),
             descending=False, nulls_first=True, limit=3, expected=[1, 2, 3]),
        dict(name=f"strings_more_nulls_than_limit_first_{st}", source=f"{S_}:3161-3169", values=arr(st, ["def", N, N, N]),
             descending=False, nulls_first=True, limit=2, expected=[1, 2]),
        dict(name=f"strings_more_nulls_than_limit_last_{st}", source=f"{S_}:3171-3179", values=arr(st, ["def", N, N, N]),
             descending=False, nulls_first=False, limit=2, expected=[0, 1]),
    ]
# test_sort_strings (:3182-3320): expected VALUES there; as indices of the stable order here (no ties in these inputs)
L1, L2 = "long string longer than 12 bytes", "lang string longer than 12 bytes"
STR8 = [N, "bad", "sad", L1, N, "glad", L2, "-ad"]
STR8B = [N, "bad", L1, "sad", N, "glad", L2, "-ad"]
sort_cases += [
    dict(name="test_sort_strings_default", source=f"{S_}:3183-3206", values=arr("Utf8", STR8), expected=[0, 4, 7, 1, 5, 6, 3, 2]),
    dict(name="test_sort_strings_desc_nulls_last", source=f"{S_}:3208-3234", values=arr("Utf8", STR8), descending=True,
         nulls_first=False, expected=[2, 3, 6, 5, 1, 7, 0, 4]),
    dict(name="test_sort_strings_asc_nulls_first", source=f"{S_}:3236-3262", values=arr("Utf8", STR8B), descending=False,
         nulls_first=True, expected=[0, 4, 7, 1, 5, 6, 2, 3]),
    dict(name="test_sort_strings_desc_nulls_first", source=f"{S_}:3264-3290", values=arr("Utf8", STR8B), descending=True,
         nulls_first=True, expected=[0, 4, 3, 2, 6, 5, 1, 7]),
    dict(name="test_sort_strings_desc_nulls_first_limit", source=f"{S_}:3292-3309", values=arr("Utf8", STR8B), descending=True,
         nulls_first=True, limit=3, expected=[0, 4, 3]),
]

# ---------------------------------------------------------------- zip (arrow-select/src/zip.rs tests)
Z_ = "arrow-select/src/zip.rs"
za = arr("Int32", [5, N, 7, N, 1])
zb = arr("Int32", [N, 3, 6, 7, 3])
m1, m2 = arr("Boolean", [T, T, F, F, T]), arr("Boolean", [F, F, T, T, F])
s42, s123, snull = arr("Int32", [42]), arr("Int32", [123]), arr("Int32", [N])
mnull = dict(type="Boolean", raw=[T, T, F, T, F, F], valid=[T, T, T, F, T, T])
zip_cases = [
    dict(name="test_zip_kernel_one", source=f"{Z_}:870", mask=m1, truthy=za, falsy=zb, expected=arr("Int32", [5, N, 6, 7, 1])),
    dict(name="test_zip_kernel_two", source=f"{Z_}:881", mask=m2, truthy=za, falsy=zb, expected=arr("Int32", [N, 3, 7, N, 3])),
    dict(name="scalar_falsy_1", source=f"{Z_}:892", mask=m1, truthy=za, falsy=s42, falsy_scalar=True, expected=arr("Int32", [5, N, 42, 42, 1])),
    dict(name="scalar_falsy_2", source=f"{Z_}:905", mask=m2, truthy=za, falsy=s42, falsy_scalar=True, expected=arr("Int32", [42, 42, 7, N, 42])),
    dict(name="scalar_truthy_1", source=f"{Z_}:918", mask=m1, truthy=s42, truthy_scalar=True, falsy=za, expected=arr("Int32", [42, 42, 7, N, 42])),
    dict(name="scalar_truthy_2", source=f"{Z_}:931", mask=m2, truthy=s42, truthy_scalar=True, falsy=za, expected=arr("Int32", [5, N, 42, 42, 1])),
    dict(name="scalar_both_ends_true", source=f"{Z_}:944", mask=m1, truthy=s42, truthy_scalar=True, falsy=s123, falsy_scalar=True,
         expected=arr("Int32", [42, 42, 123, 123, 42])),
    dict(name="scalar_both_ends_false", source=f"{Z_}:956", mask=arr("Boolean", [T, T, F, T, F, F]), truthy=s42, truthy_scalar=True,
         falsy=s123, falsy_scalar=True, expected=arr("Int32", [42, 42, 123, 42, 123, 123])),
    dict(name="scalar_none_1", source=f"{Z_}:975", mask=m1, truthy=s42, truthy_scalar=True, falsy=snull, falsy_scalar=True,
         expected=arr("Int32", [42, 42, N, N, 42])),
    dict(name="scalar_none_2", source=f"{Z_}:987", mask=m2, truthy=s42, truthy_scalar=True, falsy=snull, falsy_scalar=True,
         expected=arr("Int32", [N, N, 42, 42, N])),
    dict(name="scalar_both_null", source=f"{Z_}:999", mask=m2, truthy=snull, truthy_scalar=True, falsy=snull, falsy_scalar=True,
         expected=arr("Int32", [N, N, N, N, N])),
    dict(name="mask_nulls_are_false_arrays", source=f"{Z_}:1011", mask=mnull, truthy=arr("Int32", [1, 2, 3, 4, 5, 6]),
         falsy=arr("Int32", [7, 8, 9, 10, 11, 12]), expected=arr("Int32", [1, 2, 9, 10, 11, 12])),
    dict(name="mask_nulls_are_false_scalars", source=f"{Z_}:1038", mask=mnull, truthy=s42, truthy_scalar=True, falsy=s123,
         falsy_scalar=True, expected=arr("Int32", [42, 42, 123, 123, 123, 123])),
    dict(name="type_mismatch", source=f"{Z_}:115-119", mask=m1, truthy=za, falsy=arr("Int64", [1, 2, 3, 4, 5]),
         error="InvalidArgumentError", message="arguments need to have the same data type"),
    dict(name="length_mismatch", source=f"{Z_}:126-130", mask=m1, truthy=arr("Int32", [1, 2]), falsy=zb,
         error="InvalidArgumentError", message="all arrays should have the same length"),
    dict(name="scalar_len", source=f"{Z_}:121-125", mask=m1, truthy=arr("Int32", [1, 2]), truthy_scalar=True, falsy=zb,
         error="InvalidArgumentError", message="scalar arrays must have 1 element"),
]

# ---------------------------------------------------------------- string compare (arrow-ord/src/comparison.rs:1246-1432)
C_ = "arrow-ord/src/comparison.rs"
names4 = ["arrow", "datafusion", "flight", "parquet"]
cmp_utf8_cases = [
    dict(name="test_utf8_array_eq", source=f"{C_}:1246", op="eq", lhs=["arrow"] * 4, rhs=["arrow", "parquet", "datafusion", "flight"],
         expected=[T, F, F, F]),
    dict(name="test_utf8_array_eq_scalar", source=f"{C_}:1260", op="eq", lhs=["arrow", "parquet", "datafusion", "flight"],
         rhs_scalar="arrow", expected=[T, F, F, F]),
    dict(name="test_utf8_array_neq", source=f"{C_}:1282", op="neq", lhs=["arrow"] * 4, rhs=["arrow", "parquet", "datafusion", "flight"],
         expected=[F, T, T, T]),
    dict(name="test_utf8_array_neq_scalar", source=f"{C_}:1296", op="neq", lhs=["arrow", "parquet", "datafusion", "flight"],
         rhs_scalar="arrow", expected=[F, T, T, T]),
]
for opname, line, exp in (("lt", 1311, [T, T, F, F]), ("lt_eq", 1347, [T, T, T, F]), ("gt", 1376, [F, F, F, T]),
                          ("gt_eq", 1412, [F, F, T, T])):
    cmp_utf8_cases.append(dict(name=f"test_utf8_array_{opname}", source=f"{C_}:{line}", op=opname, lhs=names4,
                               rhs=["flight"] * 4, expected=exp))
    cmp_utf8_cases.append(dict(name=f"test_utf8_array_{opname}_scalar", source=f"{C_}:{line + 14}", op=opname, lhs=names4,
                               rhs_scalar="flight", expected=exp))
for opname, line, scalar, exp in (("eq", "1909-1918", "xyz", [F, F, T]), ("lt", "1937-1946", "xyz", [T, T, F]), ("lt_eq", "1965-1974", "def", [T, T, F]),
                                  ("gt_eq", "1993-2002", "def", [F, T, T]), ("gt", "2021-2030", "def", [F, F, T]), ("neq", "2049-2058", "xyz", [T, T, F])):
    cmp_utf8_cases.append(dict(name=f"test_{opname}_dyn_utf8_scalar", source=f"{C_}:{line}", op=opname, lhs=["abc", "def", "xyz"],
                               rhs_scalar=scalar, expected=exp))
# ---------------------------------------------------------------- like (arrow-string/src/like.rs tests, scalar patterns)
LK = "arrow-string/src/like.rs"
like_cases = []


def lk(name, values, pattern, op, expected):
    like_cases.append(dict(name=name, source=f"{LK} ({name})", values=values, pattern=pattern, op=op, expected=expected))


A5 = ["arrow", "parrow", "arrows", "arr", "arrow long string longer than 12 bytes"]
lk("test_utf8_array_like_scalar_escape_testing", ["varchar(255)", "int(255)longer than 12 bytes", "varchar", "int"], "%(%)%", "like",
   [T, T, F, F])
lk("test_utf8_array_like_scalar_escape_regex", [".*", "a", "*"], ".*", "like", [T, F, F])
lk("test_utf8_array_like_scalar_escape_regex_dot", [".", "a", "*"], ".", "like", [T, F, F])
lk("test_utf8_array_like_scalar", ["arrow", "parquet", "datafusion", "flight", "long string arrow test 12 bytes"], "%ar%", "like",
   [T, T, F, F, T])
lk("test_utf8_array_like_scalar_start", A5, "arrow%", "like", [T, F, T, F, T])
lk("test_utf8_and_binary_array_starts_with_scalar_start", A5, "arrow", "starts_with", [T, F, T, F, T])
lk("test_utf8_array_like_scalar_end", A5, "%arrow", "like", [T, T, F, F, F])
lk("test_utf8_and_binary_array_ends_with_scalar_end", A5, "arrow", "ends_with", [T, T, F, F, F])
lk("test_utf8_array_like_scalar_equals", A5, "arrow", "like", [T, F, F, F, F])
lk("test_utf8_array_like_scalar_one", ["arrow", "arrows", "parrow", "arr", "arrow long string longer than 12 bytes"], "arrow_", "like",
   [F, T, F, F, F])
lk("test_utf8_scalar_like_escape", ["a%", "a\\x", "arrow long string longer than 12 bytes"], "a\\%", "like", [T, F, F])
lk("test_utf8_scalar_like_escape_contains", ["ba%", "ba\\x", "arrow long string longer than 12 bytes"], "%a\\%", "like", [T, F, F])
lk("test_utf8_array_nlike_escape_testing", ["varchar(255)", "int(255) arrow long string longer than 12 bytes", "varchar", "int"],
   "%(%)%", "nlike", [F, F, T, T])
lk("test_utf8_array_nlike_scalar_escape_regex", [".*", "a", "*"], ".*", "nlike", [F, T, T])
lk("test_utf8_array_nlike_scalar_escape_regex_dot", [".", "a", "*"], ".", "nlike", [F, T, T])
lk("test_utf8_array_nlike_scalar", ["arrow", "parquet", "datafusion", "flight", "arrow long string longer than 12 bytes"], "%ar%",
   "nlike", [F, F, T, T, F])
lk("test_utf8_array_nlike_scalar_start", A5, "arrow%", "nlike", [F, T, F, T, F])
lk("test_utf8_array_nlike_scalar_end", A5, "%arrow", "nlike", [F, F, T, T, T])
lk("test_utf8_array_nlike_scalar_equals", A5, "arrow", "nlike", [F, T, T, T, T])
lk("test_utf8_array_nlike_scalar_one", ["arrow", "arrows", "parrow", "arr", "arrow long string longer than 12 bytes"], "arrow_", "nlike",
   [T, F, T, T, T])
MB = ["sdlkdfFooßsdfs", "sdlkdfFooSSdggs", "sdlkdfFoosssdsd", "FooS", "Foos", "ﬀooSS", "ﬀooß", "😃sadlksffofsSsh😈klF",
      "😱slgffoesSsh😈klF", "FFKoSS", "longer than 12 bytes FFKoSS"]
lk("test_uff8_array_like_multibyte", MB, "%Ssh😈klF", "like", [F, F, F, F, F, F, F, T, T, F, F])
NL = ["Earth", "Fire", "Water", "Air", N, "Air", "bbbbb\nAir"]
lk("test_utf8_scalar_nullable_like", NL, "Air", "like", [F, F, F, T, N, T, F])
lk("test_utf8_scalar_nullable_nlike", NL, "%a%r%", "nlike", [F, T, F, T, N, T, T])

# ---------------------------------------------------------------- rank (arrow-ord/src/rank.rs tests) and shift (window.rs tests)
RK = "arrow-ord/src/rank.rs"
rank_cases = []
OPTS = {"default": (F, T), "descending": (T, T), "nulls_last": (F, F), "nulls_last_descending": (T, F)}
R_PRIM = arr("Int32", [1, 1, N, 3, 3, 4])
for oname, exp in (("default", [3, 3, 1, 5, 5, 6]), ("descending", [6, 6, 1, 4, 4, 2]), ("nulls_last", [2, 2, 6, 4, 4, 5]),
                   ("nulls_last_descending", [5, 5, 6, 3, 3, 1])):
    rank_cases.append(dict(name=f"test_primitive_{oname}", source=f"{RK}:248-275", op="rank", values=R_PRIM,
                           descending=OPTS[oname][0], nulls_first=OPTS[oname][1], expected=exp))
# "Test with non-zero null values" (:277-281): the slots under the nulls hold 3, 5, 5
rank_cases.append(dict(name="test_primitive_nonzero_null_values", source=f"{RK}:277-281", op="rank",
                       values=arr("Int32", [1, 4, 3, 4, 5, 5]), validity=[T, T, F, T, F, F], descending=F, nulls_first=T,
                       expected=[4, 6, 3, 6, 3, 3]))
R_NB = arr("Boolean", [T, T, N, F, F])
for oname, exp in (("default", [5, 5, 1, 3, 3]), ("descending", [3, 3, 1, 5, 5]), ("nulls_last", [4, 4, 5, 2, 2]),
                   ("nulls_last_descending", [2, 2, 5, 4, 4])):
    rank_cases.append(dict(name=f"test_nullable_booleans_{oname}", source=f"{RK}:293-325", op="rank", values=R_NB,
                           descending=OPTS[oname][0], nulls_first=OPTS[oname][1], expected=exp))
rank_cases.append(dict(name="test_nullable_booleans_nonzero_null_values", source=f"{RK}:327-331", op="rank",
                       values=arr("Boolean", [T, T, T, F, F]), validity=[T, T, F, T, T], descending=F, nulls_first=T,
                       expected=[5, 5, 1, 3, 3]))
R_B = arr("Boolean", [T, F, F, F, T])
for oname, exp in (("default", [5, 3, 3, 3, 5]), ("descending", [2, 5, 5, 5, 2]), ("nulls_last", [5, 3, 3, 3, 5]),
                   ("nulls_last_descending", [2, 5, 5, 5, 2])):
    rank_cases.append(dict(name=f"test_booleans_{oname}", source=f"{RK}:334-362", op="rank", values=R_B,
                           descending=OPTS[oname][0], nulls_first=OPTS[oname][1], expected=exp))
for st in ("Utf8", "LargeUtf8"):
    rank_cases.append(dict(name=f"test_bytes_{st}", source=f"{RK}:365-374", op="rank", values=arr(st, ["foo", "fo", "bar", "bar"]),
                           descending=F, nulls_first=T, expected=[4, 3, 2, 2]))
rank_cases.append(dict(name="doc_example_strings", source=f"{RK}:51-56", op="rank", values=arr("Utf8", ["foo", N, "foo", N, "bar"]),
                       descending=F, nulls_first=T, expected=[5, 2, 5, 2, 3]))
# test_string_view_with_nulls (:396-405) — the same values as a Utf8 array
rank_cases.append(dict(name="test_string_with_nulls", source=f"{RK}:396-405", op="rank",
                       values=arr("Utf8", ["a string longer than twelve bytes", "bar", N, "a string longer than twelve bytes"]),
                       descending=F, nulls_first=T, expected=[3, 4, 1, 3]))
WN = "arrow-select/src/window.rs"
S3 = arr("Int32", [1, N, 4])
F3 = arr("Float64", [1.0, N, 4.0])
for name, lines, values, off, exp in (
        ("doc_shift_right_1", "37-39", S3, 1, arr("Int32", [N, 1, N])), ("doc_shift_left_1", "42-44", S3, -1, arr("Int32", [N, 4, N])),
        ("doc_shift_0", "47-49", S3, 0, arr("Int32", [1, N, 4])), ("doc_shift_right_3", "52-54", S3, 3, arr("Int32", [N, N, N])),
        ("test_shift_neg", "88-93", S3, -1, arr("Int32", [N, 4, N])), ("test_shift_pos", "96-101", S3, 1, arr("Int32", [N, 1, N])),
        ("test_shift_neg_float64", "104-109", F3, -1, arr("Float64", [N, 4.0, N])),
        ("test_shift_pos_float64", "112-117", F3, 1, arr("Float64", [N, 1.0, N])),
        ("test_shift_nil", "146-151", S3, 0, arr("Int32", [1, N, 4])),
        ("test_shift_boundary_pos", "154-159", S3, 3, arr("Int32", [N, N, N])),
        ("test_shift_boundary_neg", "162-167", S3, -3, arr("Int32", [N, N, N])),
        ("test_shift_boundary_neg_min", "170-175", S3, -2**63, arr("Int32", [N, N, N])),
        ("test_shift_large_pos", "178-183", S3, 1000, arr("Int32", [N, N, N])),
        ("test_shift_large_neg", "186-191", S3, -1000, arr("Int32", [N, N, N]))):
    rank_cases.append(dict(name=name, source=f"{WN}:{lines}", op="shift", values=values, offset=off, expected=exp))

# ---------------------------------------------------------------- temporal casts
# arrow-cast/src/cast/mod.rs tests of the "temporal casts" arms (:1700-2260).  Types are written in the reference's
# Debug form; zones are the fixed offsets the tests use.  Where a test builds its input by parsing strings (marked
# "derived"), the epoch values below were computed from those strings by hand (and re-checked with Python's datetime
# in tests/test_temporal_cast_cpu.py); where it only asserts `is_err()` / `contains(..)`, the full message is the
# format string of the cited arm.
CM = "arrow-cast/src/cast/mod.rs"
I64MAX = 2**63 - 1
tcast_cases = []


def TS(unit, tz=None):
    return f"Timestamp({unit}, None)" if tz is None else f'Timestamp({unit}, Some("{tz}"))'


def tcast(name, lines, values, to, expected=None, safe=True, **kw):
    d = dict(name=name, source=f"{CM}:{lines}", values=values, to=to, safe=safe, **kw)
    if expected is not None:
        d["expected"] = arr(to, expected)
    tcast_cases.append(d)


tcast("test_cast_date32_to_date64", "5296-5303", arr("Date32", [10000, 17890]), "Date64", [864000000000, 1545696000000])
tcast("test_cast_date64_to_date32", "5306-5314", arr("Date64", [864000000005, 1545696000001, N]), "Date32", [10000, 17890, N])
tcast("test_cast_date64_to_date32_overflow_safe", "5317-5323", arr("Date64", [I64MAX]), "Date32", [N])
tcast("test_cast_date64_to_date32_overflow_unsafe", "5325-5333", arr("Date64", [I64MAX]), "Date32", safe=F,
      error="CastError", message=f"Cannot cast Date64 value {I64MAX} to Date32 without overflow")
tcast("test_cast_date32_to_int32", "6618-6624", arr("Date32", [10000, 17890]), "Int32", [10000, 17890])
tcast("test_cast_int32_to_date32", "6627-6633", arr("Int32", [10000, 17890]), "Date32", [10000, 17890])
tcast("test_cast_timestamp_to_date32", "6636-6645", arr(TS("Millisecond", "+00:00"), [864000000005, 1545696000001, N]),
      "Date32", [10000, 17890, N])
# derived: "1970-01-01T00:00:01" / "1970-01-01T23:59:59" read in -07:00, "2020-03-01T02:00:23+00:00"
tcast("test_cast_timestamp_to_date32_zone", "6647-6665 (derived)",
      arr(TS("Millisecond", "-07:00"), [25201000, 111599000, N, 1583028023000]), "Date32", [0, 0, N, 18321])
tcast("test_cast_timestamp_to_date64_ms", "6667-6675", arr(TS("Millisecond"), [864000000005, 1545696000001, N]), "Date64",
      [864000000005, 1545696000001, N])
tcast("test_cast_timestamp_to_date64_s", "6677-6681", arr(TS("Second"), [864000000005, 1545696000001]), "Date64",
      [864000000005000, 1545696000001000])
tcast("test_cast_timestamp_to_date64_overflow_safe", "6683-6686", arr(TS("Second"), [I64MAX]), "Date64", [N])
tcast("test_cast_timestamp_to_date64_overflow_unsafe", "6687-6694", arr(TS("Second"), [I64MAX]), "Date64", safe=F,
      error="ArithmeticOverflow", message=f"Overflow happened on: {I64MAX} * 1000")
for unit, k in (("Second", 1), ("Millisecond", 10**3), ("Microsecond", 10**6), ("Nanosecond", 10**9)):
    src = arr(TS(unit, "+01:00"), [86405 * k, 1 * k, N])
    tcast(f"test_cast_timestamp_to_time64_{unit}_us", "6697-6755", src, "Time64(Microsecond)", [3605000000, 3601000000, N])
    tcast(f"test_cast_timestamp_to_time64_{unit}_ns", "6697-6755", src, "Time64(Nanosecond)", [3605000000000, 3601000000000, N])
    tcast(f"test_cast_timestamp_to_time32_{unit}_s", "6770-6830", src, "Time32(Second)", [3605, 3601, N])
    tcast(f"test_cast_timestamp_to_time32_{unit}_ms", "6770-6830", src, "Time32(Millisecond)", [3605000, 3601000, N])
for to in ("Time64(Microsecond)", "Time64(Nanosecond)", "Time32(Second)", "Time32(Millisecond)"):
    # `cast(..)` = safe mode, and still an error: the arm is try_unary in both modes
    tcast(f"test_cast_timestamp_to_{to}_overflow", "6757-6766, 6832-6840", arr(TS("Second", "+01:00"), [I64MAX]), to,
          error="CastError", message=f"Failed to create naive time with arrow_array::types::TimestampSecondType {I64MAX}")
tcast("test_cast_timestamp_to_time64_millisecond_unsupported", "6765-6766", arr(TS("Second", "+01:00"), [I64MAX]),
      "Time64(Millisecond)", error="CastError",
      message='Casting from Timestamp(s, "+01:00") to Time64(ms) not supported')
# derived: "2000-01-01T00:00:00.123456789" / "2010-01-01T00:00:00.123456789"
tcast("test_cast_timestamp_with_timezone_1", "6843-6862 (derived)",
      arr(TS("Nanosecond"), [946684800123456789, 1262304000123456789, N]), TS("Microsecond", "+0700"),
      [946659600123456, 1262278800123456, N])
tcast("test_cast_timestamp_with_timezone_2", "6864-6890 (derived)",
      arr(TS("Millisecond", "+0700"), [946684800123, 1262304000123, N]), TS("Nanosecond"),
      [946684800123000000, 1262304000123000000, N])
tcast("test_cast_timestamp_with_timezone_3", "6892-6917 (derived)",
      arr(TS("Microsecond", "+0700"), [946684800123456, 1262304000123456, N]), TS("Second", "-08:00"),
      [946684800, 1262304000, N])
D64 = arr("Date64", [864000000005, 1545696000001, N])
for unit, lines, exp in (("Second", "6919-6927", [864000000, 1545696000, N]), ("Millisecond", "6929-6940", [864000000005, 1545696000001, N]),
                         ("Microsecond", "6942-6953", [864000000005000, 1545696000001000, N]),
                         ("Nanosecond", "6955-6966", [864000000005000000, 1545696000001000000, N])):
    tcast(f"test_cast_date64_to_timestamp_{unit}", lines, D64, TS(unit), exp)
tcast("test_cast_timestamp_to_i64", "6968-6978", arr(TS("Millisecond", "+00:00"), [864000000005, 1545696000001, N]), "Int64",
      [864000000005, 1545696000001, N])
D32 = arr("Date32", [18628, 18993, N])
for unit, k, l32, l64 in (("Second", 1, "7067-7086", "7151-7170"), ("Millisecond", 10**3, "7088-7107", "7172-7191"),
                          ("Microsecond", 10**6, "7109-7128", "7193-7212"), ("Nanosecond", 10**9, "7130-7149", "7214-7233")):
    tcast(f"test_cast_date32_to_timestamp_with_timezone_{unit}", l32, D32, TS(unit, "+0545"), [1609438500 * k, 1640974500 * k, N])
    tcast(f"test_cast_date64_to_timestamp_with_timezone_{unit}", l64, D64, TS(unit, "+0545"),
          [863979300 * k + 5 * (k // 1000), 1545675300 * k + 1 * (k // 1000), N])
tcast("test_cast_between_timestamps", "7323-7331", arr(TS("Millisecond"), [864000003005, 1545696002001, N]), TS("Second"),
      [864000003, 1545696002, N])
for u in ("Nanosecond", "Microsecond", "Millisecond", "Second"):
    tcast(f"test_cast_duration_to_i64_{u}", "7334-7350", arr(f"Duration({u})", [5, 6, 7, 8, 100000000]), "Int64", [5, 6, 7, 8, 100000000])
UM = {"Second": 1, "Millisecond": 10**3, "Microsecond": 10**6, "Nanosecond": 10**9}
for fu in UM:
    for tu in UM:
        if fu == tu:
            continue
        v1, v2 = 8640003005, 1696002001
        if UM[fu] >= UM[tu]:
            e = [v1 // (UM[fu] // UM[tu]), v2 // (UM[fu] // UM[tu]), N]
        else:
            e = [v1 * (UM[tu] // UM[fu]), v2 * (UM[tu] // UM[fu]), N]
        tcast(f"test_cast_between_durations_{fu}_{tu}", "7353-7404", arr(f"Duration({fu})", [v1, v2, N]), f"Duration({tu})", e)
DS = arr("Duration(Second)", [I64MAX, 8640203410378005, 10241096, N])
tcast("test_cast_duration_overflow_to_null", "7406-7417", DS, "Duration(Nanosecond)", [N, N, 10241096000000000, N])
tcast("test_cast_duration_to_int64", "7419-7431", DS, "Int64", [I64MAX, 8640203410378005, 10241096, N])
tcast("test_cast_duration_to_int32", "7432-7437", DS, "Int32", [N, N, 10241096, N])
for unit, k, lines in (("Second", 1, "11680-11688"), ("Millisecond", 10**3, "11691-11702"), ("Microsecond", 10**6, "11705-11716"),
                       ("Nanosecond", 10**9, "11719-11730")):
    tcast(f"test_cast_date32_to_timestamp_{unit}", lines, D32, TS(unit), [1609459200 * k, 1640995200 * k, N])
MDM = I64MAX // 86_400_000_000
tcast("test_cast_date32_to_timestamp_us_overflow_safe", "11733-11752", arr("Date32", [MDM, MDM + 1, N]), TS("Microsecond"),
      [MDM * 86_400_000_000, N, N])
tcast("test_cast_date32_to_timestamp_us_overflow_unsafe", "11733-11746", arr("Date32", [MDM, MDM + 1, N]), TS("Microsecond"), safe=F,
      error="ArithmeticOverflow", message=f"Overflow happened on: {MDM + 1} * 86400000000")
tcast("test_cast_date32_to_timestamp_ns_overflow_safe", "11755-11775", arr("Date32", [106751, 106752, N]), TS("Nanosecond"),
      [106751 * 86_400_000_000_000, N, N])
tcast("test_cast_date32_to_timestamp_ns_overflow_unsafe", "11755-11769", arr("Date32", [106751, 106752, N]), TS("Nanosecond"), safe=F,
      error="ArithmeticOverflow", message="Overflow happened on: 106752 * 86400000000000")
# derived: "1900-01-03 23:59:59", "1969-12-31 00:00:01", "1989-12-31 00:00:01" -> ns / 1_000_000 (truncating)
tcast("test_cast_below_unixtimestamp", "12418-12456 (derived)",
      arr(TS("Millisecond", "+00:00"), [-2208729601000, -86399000, 631065601000]), "Date32", [-25565, -1, 7304])
tcast("test_cast_time32_second_to_int64", "14056-14077", arr("Time32(Second)", [1000, 2000, 3000]), "Int64", [1000, 2000, 3000])
tcast("test_cast_time32_millisecond_to_int64", "14080-14101", arr("Time32(Millisecond)", [1000, 2000, 3000]), "Int64", [1000, 2000, 3000])
tcast("test_cast_time32_millisecond_to_time64_nanosecond", "14104-14113", arr("Time32(Millisecond)", [1000, 2000, N, 43200000]),
      "Time64(Nanosecond)", [1000000000, 2000000000, N, 43200000000000])
tcast("test_cast_time32_millisecond_to_time64_microsecond", "14116-14125", arr("Time32(Millisecond)", [1000, 2000, N, 43200000]),
      "Time64(Microsecond)", [1000000, 2000000, N, 43200000000])
tcast("test_cast_time32_second_to_time64_nanosecond", "14128-14136", arr("Time32(Second)", [1, 60, N, 43200]),
      "Time64(Nanosecond)", [1000000000, 60000000000, N, 43200000000000])
tcast("test_cast_time32_second_to_time64_microsecond", "14139-14147", arr("Time32(Second)", [1, 60, N, 43200]),
      "Time64(Microsecond)", [1000000, 60000000, N, 43200000000])
tcast("test_cast_time32_second_to_time32_millisecond_overflow_safe", "14150-14155", arr("Time32(Second)", [2**31 - 1]),
      "Time32(Millisecond)", [N])
tcast("test_cast_time32_second_to_time32_millisecond_overflow_unsafe", "14156-14164", arr("Time32(Second)", [2**31 - 1]),
      "Time32(Millisecond)", safe=F, error="ArithmeticOverflow", message="Overflow happened on: 2147483647 * 1000")
for t in ("Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64"):
    tcast(f"test_cast_integer_to_timestamp_{t}", "5056-5094", arr(t, [2, 10, N]), TS("Microsecond"), [2, 10, N])
    tcast(f"test_cast_timestamp_to_integer_{t}", "5097-5123", arr(TS("Millisecond", "+00:00"), [5, 1, N]), t, [5, 1, N])


# ------------------------------------------------------------- temporal arithmetic
# arrow-arith/src/numeric.rs tests with temporal operands (no intervals): ops as AH_ADD=0, AH_ADD_WRAPPING=1, AH_SUB=2,
# AH_SUB_WRAPPING=3, AH_MUL=4, AH_MUL_WRAPPING=5, AH_DIV=6, AH_REM=7.
NM = "arrow-arith/src/numeric.rs"
tarith_cases = []


def tarith(name, lines, op, lhs, rhs, expected=None, **kw):
    d = dict(name=name, source=f"{NM}:{lines}", op=op, lhs=lhs, rhs=rhs, **kw)
    if expected is not None:
        d["expected"] = expected
    tarith_cases.append(d)


for u in ("Second", "Millisecond", "Microsecond", "Nanosecond"):
    a = arr(TS(u), [2000000, 434030324, 53943340])
    b = arr(TS(u), [329593, 59349, 694994])
    dur = arr(f"Duration({u})", [1670407, 433970975, 53248346])
    tarith(f"test_timestamp_sub_{u}", "1548-1556", 2, a, b, dur)
    tarith(f"test_timestamp_add_duration_{u}", "1558-1559", 0, b, dur, a)
    tarith(f"test_duration_add_timestamp_{u}", "1561-1562", 0, dur, b, a)
    da, db = arr(f"Duration({u})", [1000, 4394, -3944]), arr(f"Duration({u})", [4, -5, -243])
    tarith(f"test_duration_add_{u}", "1972-1977", 0, da, db, arr(f"Duration({u})", [1004, 4389, -4187]))
    tarith(f"test_duration_sub_{u}", "1978-1979", 2, da, db, arr(f"Duration({u})", [996, 4399, -3701]))
    sym = {4: "*", 6: "/", 7: "%"}
    for op in (4, 6, 7):
        us = {"Second": "s", "Millisecond": "ms", "Microsecond": "µs", "Nanosecond": "ns"}[u]
        tarith(f"test_duration_invalid_{op}_{u}", "1981-1997", op, da, db, error="InvalidArgumentError",
               message=f"Invalid duration arithmetic operation: Duration({us}) {sym[op]} Duration({us})")
    tarith(f"test_duration_overflow_{u}", "1999-2006", 0, arr(f"Duration({u})", [I64MAX]), arr(f"Duration({u})", [1]),
           error="ArithmeticOverflow", message=f"Overflow happened on: {I64MAX} + 1",
           display=f"Arithmetic overflow: Overflow happened on: {I64MAX} + 1")
tarith("test_date32_sub", "2110-2116", 2, arr("Date32", [-2**31, 2**31 - 1, 23, 7684]), arr("Date32", [-2**31, -2**31, -2, 45]),
       arr("Duration(Second)", [0, 371085174288000, 2160000, 660009600]))
tarith("test_date64_sub", "2118-2124", 2, arr("Date64", [4343, 76676, 3434]), arr("Date64", [3, -5, 5]),
       arr("Duration(Millisecond)", [4340, 76681, 3429]))
tarith("test_date64_sub_overflow", "2126-2133", 2, arr("Date64", [I64MAX]), arr("Date64", [-1]), error="ArithmeticOverflow",
       message=f"Overflow happened on: {I64MAX} - -1", display=f"Arithmetic overflow: Overflow happened on: {I64MAX} - -1")


# test_decimal (numeric.rs:1400-1481) and test_decimal256_same_scale_add_sub's Decimal128 analogue: ops 0 add, 2 sub, 4 mul, 6 div, 7 rem
DA = arr("Decimal128(12, 3)", [15, 0, -577, 334, -78, 3])
DB = arr("Decimal128(12, 1)", [54, 34, -356, 3, 6, 745])
for name, op, t, exp in (("add", 0, "Decimal128(15, 3)", [5415, 3400, -36177, 634, 522, 74503]),
                         ("sub", 2, "Decimal128(15, 3)", [-5385, -3400, 35023, 34, -678, -74497]),
                         ("mul", 4, "Decimal128(25, 4)", [810, 0, 205412, 1002, -468, 2235]),
                         ("div", 6, "Decimal128(17, 7)", [27777, 0, 162078, 11133333, -1300000, 402]),
                         ("rem", 7, "Decimal128(12, 3)", [15, 0, -577, 34, -78, 3])):
    tarith(f"test_decimal_{name}", "1400-1440", op, DA, DB, arr(t, exp))
tarith("test_decimal_mul_scale_exceeds", "1442-1452", 4, arr("Decimal128(3, 3)", [1]), arr("Decimal128(37, 37)", [1]),
       error="InvalidArgumentError", message="Output scale of Decimal128(3, 3) * Decimal128(37, 37) would exceed max scale of 38",
       display="Invalid argument error: Output scale of Decimal128(3, 3) * Decimal128(37, 37) would exceed max scale of 38")
tarith("test_decimal_pow_overflow", "1454-1458", 0, arr("Decimal128(3, -2)", [1]), arr("Decimal128(37, 37)", [1]),
       error="ArithmeticOverflow", message="Overflow happened on: 10 ^ 39", display="Arithmetic overflow: Overflow happened on: 10 ^ 39")
tarith("test_decimal_scale_mul_overflow", "1460-1467", 0, arr("Decimal128(3, -1)", [10]), arr("Decimal128(37, 37)", [1]),
       error="ArithmeticOverflow", message="Overflow happened on: 10 * 100000000000000000000000000000000000000",
       display="Arithmetic overflow: Overflow happened on: 10 * 100000000000000000000000000000000000000")
tarith("test_decimal_div_by_zero", "1469-1473", 6, arr("Decimal128(3, -1)", [10]), arr("Decimal128(1, 1)", [0]),
       error="DivideByZero", message="Divide by zero error", display="Divide by zero error")
tarith("test_decimal_rem_by_zero", "1474-1475", 7, arr("Decimal128(3, -1)", [10]), arr("Decimal128(1, 1)", [0]),
       error="DivideByZero", message="Divide by zero error", display="Divide by zero error")


for name, cases in [("like", like_cases), ("cmp_utf8", cmp_utf8_cases), ("zip", zip_cases), ("sort", sort_cases), ("concat", concat_cases), ("aggregate", agg_cases), ("boolean", bool_cases), ("filter", filter_cases), ("take", take_cases), ("arith", arith_cases), ("cmp", cmp_cases),
                    ("cast", cast_cases), ("rank_shift", rank_cases), ("cast_temporal", tcast_cases), ("arith_temporal", tarith_cases)]:
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump({"reference": "apache/arrow-rs 59.2.0", "cases": cases}, f, indent=1)
    print(name, len(cases))
