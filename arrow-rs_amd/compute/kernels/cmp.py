"""arrow::compute::kernels::cmp == arrow_ord::cmp (arrow-ord/src/cmp.rs:79-202).
Floats compare in IEEE totalOrder; equality is bitwise (arrow-array/src/arithmetic.rs:400-410)."""
import ctypes as C

from ... import _lib as L
from ...array import Array, Boolean

EQ, NEQ, LT, LT_EQ, GT, GT_EQ, DISTINCT, NOT_DISTINCT = range(8)


def _compare(op, lhs, rhs):
    l, l_s = lhs.get()
    r, r_s = rhs.get()
    ctx = l.ctx
    out = L.ArrayOut()
    lv, rv = l.view(), r.view()
    if l.data_type.logical is not None or r.data_type.logical is not None:
        # compare_op's rule that the LOGICAL types agree (cmp.rs:243-264: Decimal128(12, 3) vs Decimal128(12, 1) is refused)
        # lives behind the C ABI (ah_compare_with_types), not here: a Rust host gets it from the same place
        lt, rt = l.data_type.descriptor(), r.data_type.descriptor()
        ctx.check(ctx.lib.ah_compare_with_types(ctx.handle, op, C.byref(lv), int(l_s), C.byref(lt), C.byref(rv), int(r_s),
                                                C.byref(rt), C.byref(out)))
        return Array._from_out(ctx, out, Boolean)
    ctx.check(ctx.lib.ah_compare(ctx.handle, op, C.byref(lv), int(l_s), C.byref(rv), int(r_s),
                                 C.byref(out)))
    return Array._from_out(ctx, out, Boolean)


def eq(lhs, rhs):
    """cmp.rs:79"""
    return _compare(EQ, lhs, rhs)


def neq(lhs, rhs):
    """cmp.rs:96"""
    return _compare(NEQ, lhs, rhs)


def lt(lhs, rhs):
    """cmp.rs:113"""
    return _compare(LT, lhs, rhs)


def lt_eq(lhs, rhs):
    """cmp.rs:130"""
    return _compare(LT_EQ, lhs, rhs)


def gt(lhs, rhs):
    """cmp.rs:147"""
    return _compare(GT, lhs, rhs)


def gt_eq(lhs, rhs):
    """cmp.rs:164"""
    return _compare(GT_EQ, lhs, rhs)


def distinct(lhs, rhs):
    """cmp.rs:182"""
    return _compare(DISTINCT, lhs, rhs)


def not_distinct(lhs, rhs):
    """cmp.rs:200"""
    return _compare(NOT_DISTINCT, lhs, rhs)
