"""String predicates on device (arrow_string::like / length; arrow-string/src/like.rs, predicate.rs, length.rs):
the HIP kernels, through the C ABI, against the reference's goldens and the CPU oracle (which is itself pinned to a
real regex engine in tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, assert_logical_eq, load_golden

pytestmark = pytest.mark.gpu

FN = {"like": K.like, "nlike": K.nlike, "starts_with": K.starts_with, "ends_with": K.ends_with, "contains": K.contains}


def host(a):
    return HostArray.from_device(a)


@pytest.mark.parametrize("case", load_golden("like"), ids=lambda c: c["name"])
@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=str)
def test_like_reference_goldens(ctx, case, dt):
    v = HostArray.from_pylist(case["values"], dt)
    got = FN[case["op"]](v.to_device(ctx), case["pattern"])
    assert_logical_eq(host(got), HostArray.from_pylist(case["expected"], A.Boolean), case["name"])
    assert (got.validity is None) == (v.valid is None)
    assert got.null_count() == sum(x is None for x in case["values"])


@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=str)
def test_like_fuzz_vs_oracle(ctx, oracle, dt):
    rng = np.random.default_rng(21)
    sym = ["a", "b", "c", "%", "_", "\\", ".", "*", "ß", "😈", "\n", "ab", "abc"]
    for it in range(10):
        n = int(rng.integers(1, 1500))
        rows = ["".join(rng.choice(sym, int(rng.integers(0, 12)))) for _ in range(n)]
        valid = (rng.random(n) < 0.85) if it % 2 == 0 else None
        hv = HostArray(dt, rows, valid)
        dv = hv.to_device(ctx)
        off = int(rng.integers(0, min(n, 70)))
        for _ in range(3):
            pat = "".join(rng.choice(sym, int(rng.integers(0, 7))))
            for op in FN:
                exp = oracle.string_like(op, hv, pat)
                got = FN[op](dv, pat)
                assert_logical_eq(host(got), exp, f"{op} {pat!r} iter {it}")
                assert (got.validity is None) == (exp.valid is None)
                # value bits under null slots are computed too (from_unary): compare the raw bits
                assert np.array_equal(np.asarray(host(got).values), np.asarray(exp.values)), f"{op} {pat!r} raw bits"
            sl = FN["like"](dv.slice(off, n - off), pat)
            assert_logical_eq(host(sl), oracle.string_like("like", hv.slice(off, n - off), pat), f"sliced {pat!r}")


def test_like_long_rows_and_backtracking(ctx, oracle):
    rows = ["a" * 300 + "b", "a" * 300, "ab" * 200, "", "x" + "😈" * 100 + "y", "a" * 50 + "%" + "a" * 50]
    hv = HostArray.from_pylist(rows, A.LargeUtf8)
    dv = hv.to_device(ctx)
    for pat in ["%a%a%a%b", "a%a%a%a%a%c", "%" + "_" * 100 + "%", "x%y", "%😈_y", "a" * 50 + "\\%%", "%%%", "", "_" * 301,
                "%" + "ab" * 30 + "%", "a" * 600]:
        for op in ("like", "nlike"):
            assert_logical_eq(host(FN[op](dv, pat)), oracle.string_like(op, hv, pat), f"{op} {pat[:20]!r}")
    long_pat = "%" + "ab_" * 250 + "%"  # a token program longer than the LDS copy (512 tokens)
    assert_logical_eq(host(K.like(dv, long_pat)), oracle.string_like("like", hv, long_pat), "long pattern")


def test_like_null_scalar_errors_and_empty(ctx):
    dv = HostArray.from_pylist(["a", None, "b"], A.Utf8).to_device(ctx)
    null_pat = A.Scalar(A.Array.from_strings([""], np.array([False]), A.Utf8, ctx))
    r = K.like(dv, null_pat)
    assert r.to_pylist() == [None, None, None] and r.null_count() == 3          # like.rs:314
    with pytest.raises(A.array.InvalidArgumentError) as ei:                       # like.rs:290
        K.like(dv, A.Scalar(A.Array.from_strings(["a"], None, A.LargeUtf8, ctx)))
    assert ei.value.message == "Invalid string/binary operation: Utf8 LIKE LargeUtf8"
    with pytest.raises(A.array.InvalidArgumentError) as ei:                       # like.rs:223
        K.like(dv, A.Array.from_strings(["a", "b"], None, A.Utf8, ctx))
    assert ei.value.message == "Cannot compare arrays of different lengths, got 3 vs 2"
    with pytest.raises(A.array.NotYetImplemented):
        K.like(dv, A.Array.from_strings(["a", "b", "c"], None, A.Utf8, ctx))
    empty = K.like(A.Array.from_strings([], None, A.Utf8, ctx), "%")
    assert empty.length == 0 and empty.to_pylist() == []


@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=str)
def test_length_and_bit_length(ctx, oracle, dt):
    rng = np.random.default_rng(22)
    rows = ["x" * int(rng.integers(0, 40)) + ("ß😈" if i % 3 == 0 else "") for i in range(5000)]
    hv = HostArray(dt, rows, rng.random(5000) < 0.9)
    dv = hv.to_device(ctx)
    assert_logical_eq(host(K.length(dv)), oracle.string_length(hv), "length")
    assert_logical_eq(host(K.bit_length(dv)), oracle.string_length(hv, bits=True), "bit_length")
    assert_logical_eq(host(K.length(dv.slice(17, 1000))), oracle.string_length(hv.slice(17, 1000)), "sliced length")
    # length.rs tests: "hello", " ", "world", null
    small = A.Array.from_strings(["hello", " ", "world", ""], np.array([True, True, True, False]), dt, ctx)
    assert K.length(small).to_pylist() == [5, 1, 5, None] and K.bit_length(small).to_pylist() == [40, 8, 40, None]


def test_like_feeds_filter_without_leaving_the_device(ctx, oracle):
    """`like(col, pattern) -> and(is_not_null) -> filter(col, ..)`: the string flavour of predicate construction."""
    rng = np.random.default_rng(23)
    n = 50_000
    words = ["arrow", "parquet", "datafusion", "flight", "barrow", "tar", ""]
    rows = [words[i] + str(j) for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 50, n))]
    hv = HostArray(A.Utf8, rows, rng.random(n) < 0.9)
    dv = hv.to_device(ctx)
    pred = K.like(dv, "%ar%1_")
    got = K.filter(dv, pred)
    exp = oracle.filter(hv, oracle.string_like("like", hv, "%ar%1_"))
    assert_logical_eq(host(got), exp, "like -> filter")
    assert 0 < got.length < n // 4
