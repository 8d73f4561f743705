"""The reference's own RowSelection unit tests, transcribed (parquet/src/arrow/arrow_reader/selection/*.rs; each
case cites its source).  `S` is a constructor namespace: tests/selection_model.py's ModelSelection (oracle, CPU)
or the DeviceAdapter around arrow_rs_amd.selection.RowSelection (HIP)."""
import pytest

import arrow_rs_amd as A
from arrow_rs_amd.selection import RowSelector
from selection_model import bools_of

sel, skip = RowSelector.select, RowSelector.skip_rows
T, F = True, False


def case_test_and(S):  # algebra.rs:441-466
    a = S.from_selectors([skip(12), sel(23), skip(3), sel(5)])
    b = S.from_selectors([sel(5), skip(4), sel(15), skip(4)])
    expected = S.from_selectors([skip(12), sel(5), skip(4), sel(14), skip(3), sel(1), skip(4)])
    assert a.and_then(b) == expected
    a.split_off(7)
    expected.split_off(7)
    assert a.and_then(b) == expected
    a = S.from_selectors([sel(5), skip(3)])
    b = S.from_selectors([sel(2), skip(1), sel(1), skip(1)])
    assert a.and_then(b).selectors() == [sel(2), skip(1), sel(1), skip(4)]


def case_test_and_longer_shorter(S):  # algebra.rs:489-512
    a = S.from_selectors([sel(3), skip(33), sel(3), skip(33)])
    with pytest.raises(A.Panic, match="selection exceeds the number of selected rows"):
        a.and_then(S.from_selectors([sel(36)]))
    with pytest.raises(A.Panic, match="selection contains less than the number of selected rows"):
        a.and_then(S.from_selectors([sel(3)]))


def case_test_intersect_and_combine(S):  # algebra.rs:515-580
    r = S.from_selectors([sel(5), skip(4), sel(1)]).intersection(S.from_selectors([sel(8), skip(1), sel(1)]))
    assert r.selectors() == [sel(5), skip(4), sel(1)]
    r = S.from_selectors([sel(3), skip(33), sel(3), skip(33)]).intersection(S.from_selectors([sel(36), skip(36)]))
    assert r.selectors() == [sel(3), skip(69)]
    r = S.from_selectors([sel(3), skip(7)]).intersection(S.from_selectors([sel(2), skip(2), sel(2), skip(2), sel(2)]))
    assert r.selectors() == [sel(2), skip(8)]


def case_test_intersection(S):  # algebra.rs:615-642
    s = S.from_selectors([sel(1048576)])
    assert s.intersection(s) == s
    a = S.from_selectors([skip(10), sel(10), skip(10), sel(20)])
    b = S.from_selectors([skip(20), sel(20), skip(10)])
    assert a.intersection(b).selectors() == [skip(30), sel(10), skip(10)]


def case_test_union(S):  # algebra.rs:645-677
    s = S.from_selectors([sel(1048576)])
    assert s.union(s) == s
    a = S.from_selectors([skip(10), sel(10), skip(10), sel(20)])
    b = S.from_selectors([skip(20), sel(20), skip(10), sel(10), skip(10)])
    assert a.union(b).selectors() == [skip(10), sel(50), skip(10)]


def case_mask_and_then(S):  # algebra.rs:680-757
    outer = S.from_boolean_buffer([F, T, T, F, T, F, T])
    inner = S.from_filters([[T, F, T, F]])
    assert bools_of(outer.and_then(inner)) == [F, T, F, F, T, F, F]
    outer = S.from_boolean_buffer([F, T, T, F, T, F, T, T])
    inner = S.from_boolean_buffer([F, T, F, T, F])
    assert bools_of(outer.and_then(inner)) == [F, F, T, F, F, F, T, F]
    outer = S.from_filters([[F, T, T, F, T]])
    assert outer.and_then(S.from_boolean_buffer([T, F, T])) == S.from_filters([[F, T, F, F, T]])
    r = S.from_boolean_buffer([F, T, T, F, T]).and_then(S.from_boolean_buffer([F, F, F]))
    assert r.total_row_count() == 5 and r.row_count() == 0
    r = S.from_boolean_buffer([F, T, T, F, T]).and_then(S.from_boolean_buffer([T, T, T]))  # all selected: the input
    assert bools_of(r) == [F, T, T, F, T]


def case_from_selectors_normalises(S):  # selector.rs:153-165, :419-435
    assert S.from_selectors([sel(0), skip(0), sel(2), sel(0), skip(1)]).selectors() == [sel(2), skip(1)]
    assert S.from_selectors([skip(0), sel(2), skip(0), sel(2)]).selectors() == [sel(4)]
    assert S.from_selectors([sel(0), skip(2), sel(0), skip(2)]).selectors() == [skip(4)]


def case_test_split_off(S):  # selector.rs:168-214
    s = S.from_selectors([skip(34), sel(12), skip(3), sel(35)])
    assert s.split_off(34).selectors() == [skip(34)]
    assert s.selectors() == [sel(12), skip(3), sel(35)]
    assert s.split_off(5).selectors() == [sel(5)]
    assert s.selectors() == [sel(7), skip(3), sel(35)]
    assert s.split_off(8).selectors() == [sel(7), skip(1)]
    assert s.selectors() == [skip(2), sel(35)]
    assert s.split_off(200).selectors() == [skip(2), sel(35)]
    assert s.selectors() == []


def case_test_offset(S):  # selector.rs:217-272
    s = S.from_selectors([sel(5), skip(23), sel(7), skip(33), sel(6)]).offset(2)
    assert s.selectors() == [skip(2), sel(3), skip(23), sel(7), skip(33), sel(6)]
    s = s.offset(5)
    assert s.selectors() == [skip(30), sel(5), skip(33), sel(6)]
    s = s.offset(3)
    assert s.selectors() == [skip(33), sel(2), skip(33), sel(6)]
    s = s.offset(2)
    assert s.selectors() == [skip(68), sel(6)]
    s = s.offset(3)
    assert s.selectors() == [skip(71), sel(3)]
    assert s.offset(0) is s  # mod.rs:785
    assert s.offset(3).total_row_count() == 0  # boolean.rs:694 offset exceeds selected -> empty


def case_test_limit(S):  # selector.rs:344-393
    assert S.from_selectors([sel(10), skip(90)]).limit(10) == S.from_selectors([sel(10)])
    s = S.from_selectors([sel(10), skip(10), sel(10), skip(10), sel(10)])
    assert s.limit(5).selectors() == [sel(5)]
    assert s.limit(15).selectors() == [sel(10), skip(10), sel(5)]
    assert s.limit(0).selectors() == []
    full = [sel(10), skip(10), sel(10), skip(10), sel(10)]
    assert s.limit(30).selectors() == full
    assert s.limit(100).selectors() == full


def case_test_from_ranges(S):  # selector.rs:396-416
    s = S.from_consecutive_ranges([(1, 3), (4, 6), (6, 6), (8, 8), (9, 10)], 10)
    assert s.selectors() == [skip(1), sel(2), skip(1), sel(2), skip(3), sel(1)]
    with pytest.raises(A.Panic):
        S.from_consecutive_ranges([(1, 3), (8, 10), (4, 7)], 10)


def case_test_trim(S):  # selector.rs:438-464, boolean.rs:719
    full = [skip(34), sel(12), skip(3), sel(35)]
    assert S.from_selectors(full).trim().selectors() == full
    assert S.from_selectors([skip(34), sel(12), skip(3)]).trim().selectors() == [skip(34), sel(12)]
    assert S.from_selectors([skip(20)]).trim().total_row_count() == 0


def case_test_from_filters(S):  # mod.rs:793-836
    filters = [[F, F, F, T, T, T, T], [T, T, F, F, T, T, T], [F, F, F, F], []]
    s = S.from_filters(filters[:1])
    assert s.selects_any() and s.selectors() == [skip(3), sel(4)]
    s = S.from_filters(filters[:2])
    assert s.selects_any() and s.selectors() == [skip(3), sel(6), skip(2), sel(3)]
    s = S.from_filters(filters)
    assert s.selects_any() and s.selectors() == [skip(3), sel(6), skip(2), sel(3), skip(4)]
    s = S.from_filters(filters[2:3])
    assert not s.selects_any() and s.selectors() == [skip(4)]


def case_test_counts(S):  # mod.rs:768-783, :856-881, :839-853
    s = S.from_selectors([skip(34), sel(12), skip(3), sel(35)])
    assert (s.row_count(), s.skipped_row_count(), s.total_row_count()) == (47, 37, 84)
    s = S.from_selectors([sel(12), sel(35)])
    assert (s.row_count(), s.skipped_row_count()) == (47, 0)
    s = S.from_selectors([skip(34), skip(3)])
    assert (s.row_count(), s.skipped_row_count()) == (0, 37)
    rt = [sel(3), skip(33), sel(4)]
    assert S.from_selectors(rt).selectors() == rt


def case_mask_limit_and_split(S):  # boolean.rs:651-716
    s = S.from_boolean_buffer([T, F, T, T, F, T, F])
    assert bools_of(s.limit(2)) == [T, F, T]
    assert bools_of(s.limit(10)) == [T, F, T, T, F, T, F]
    head = s.split_off(3)
    assert bools_of(head) == [T, F, T] and bools_of(s) == [T, F, T, F]
    whole = s.split_off(99)
    assert bools_of(whole) == [T, F, T, F] and s.total_row_count() == 0


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
