"""arrow::compute::kernels::aggregate == arrow_arith::aggregate (arrow-arith/src/aggregate.rs).
Every function returns a numpy scalar of the array's native type, or ``None`` where the reference
returns ``None`` (empty or all-null input).  ``sum``/``min``/``max`` shadow the Python builtins on
purpose, exactly as the reference's names do."""
import ctypes as C

import numpy as np

from ... import _lib as L

SUM, SUM_CHECKED, PRODUCT, PRODUCT_CHECKED, MIN, MAX, BIT_AND, BIT_OR, BIT_XOR = range(9)


def _aggregate(op, array):
    ctx = array.ctx
    out = L.Scalar()
    v = array.view()
    ctx.check(ctx.lib.ah_aggregate(ctx.handle, op, C.byref(v), C.byref(out)))
    if not out.is_valid:
        return None
    if array.data_type.physical == L.AH_BOOL:
        return bool(out.bytes[0])
    return np.frombuffer(bytes(out.bytes), dtype=array.data_type.np_dtype, count=1)[0]


def sum(array):  # noqa: A001
    """aggregate.rs:943 — wrapping for integers"""
    return _aggregate(SUM, array)


def sum_checked(array):
    """aggregate.rs:897 — ArithmeticOverflow as soon as a prefix of the valid values overflows"""
    return _aggregate(SUM_CHECKED, array)


def product(array):
    """aggregate.rs:953"""
    return _aggregate(PRODUCT, array)


def product_checked(array):
    """aggregate.rs:963"""
    return _aggregate(PRODUCT_CHECKED, array)


def min(array):  # noqa: A001
    """aggregate.rs:1012 — total order: NaN is greater than every other value"""
    return _aggregate(MIN, array)


def max(array):  # noqa: A001
    """aggregate.rs:1027"""
    return _aggregate(MAX, array)


def bit_and(array):
    """aggregate.rs:848"""
    return _aggregate(BIT_AND, array)


def bit_or(array):
    """aggregate.rs:855"""
    return _aggregate(BIT_OR, array)


def bit_xor(array):
    """aggregate.rs:862"""
    return _aggregate(BIT_XOR, array)


def min_boolean(array):
    """aggregate.rs:372"""
    return _aggregate(MIN, array)


def max_boolean(array):
    """aggregate.rs:430"""
    return _aggregate(MAX, array)


def bool_and(array):
    """aggregate.rs:880"""
    return min_boolean(array)


def bool_or(array):
    """aggregate.rs:887"""
    return max_boolean(array)
