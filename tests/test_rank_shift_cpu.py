"""rank (arrow-ord/src/rank.rs) and shift (arrow-select/src/window.rs): the oracle against the reference's own test
literals (tests/golden/rank_shift.json) and against an independent numpy / scipy formulation."""
import numpy as np
import pytest
from scipy.stats import rankdata

import arrow_rs_amd as A
from orc import HostArray, assert_logical_eq, golden_array, load_golden


def golden_values(case):
    v = golden_array(case["values"])
    if "validity" in case:  # explicit validity over non-default slot values
        v = HostArray(v.data_type, v.values, np.array(case["validity"], dtype=bool))
    return v


@pytest.mark.parametrize("case", load_golden("rank_shift"), ids=lambda c: c["name"])
def test_rank_shift_golden(oracle, case):
    v = golden_values(case)
    if case["op"] == "rank":
        assert oracle.rank(v, case["descending"], case["nulls_first"]).tolist() == case["expected"]
    else:
        assert_logical_eq(oracle.shift(v, case["offset"]), golden_array(case["expected"]), case["name"])


@pytest.mark.parametrize("dt", [A.Int8, A.Int64, A.UInt16, A.Float32, A.Float64], ids=str)
def test_rank_vs_scipy(oracle, dt):
    """`rankdata(method="max")` is the same definition (ties share the highest position) for total orders."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 17, 1000):
        vals = rng.integers(-5, 6, n).astype(dt.np_dtype) if np.dtype(dt.np_dtype).kind != "u" else rng.integers(0, 9, n).astype(dt.np_dtype)
        valid = rng.random(n) < 0.8
        hv = HostArray(dt, vals, valid)
        nv, nn = int(valid.sum()), int(n - valid.sum())
        for desc in (False, True):
            for nf in (True, False):
                exp = np.zeros(n, dtype=np.int64)
                if nv:
                    key = vals[valid].astype(np.float64)
                    exp[valid] = rankdata(-key if desc else key, method="max").astype(np.int64) + (nn if nf else 0)
                exp[~valid] = nn if nf else n
                assert oracle.rank(hv, desc, nf).tolist() == exp.tolist(), (dt, n, desc, nf)


def test_rank_float_total_order(oracle):
    """is_eq / compare on floats are totalOrder: -0.0 < 0.0, NaNs sort by sign and payload and equal only bitwise."""
    nan2 = np.frombuffer(np.uint64(0x7FF8000000000001).tobytes(), dtype=np.float64)[0]
    vals = np.array([0.0, -0.0, np.nan, -np.nan, nan2, np.inf, -np.inf, 0.0, np.nan], dtype=np.float64)
    got = oracle.rank(HostArray(A.Float64, vals)).tolist()
    # order: -nan < -inf < -0.0 < 0.0 (x2) < inf < nan (x2) < nan2
    assert got == [5, 3, 8, 1, 9, 6, 2, 5, 8]
