"""arrow::compute::kernels::numeric == arrow_arith::numeric (arrow-arith/src/numeric.rs:36-186).
All binary kernels take two ``Datum`` (``Array`` or ``Scalar``).  When an operand is a temporal logical type the
result type follows ``arithmetic_op`` / ``timestamp_op`` / ``duration_op`` / ``date_op`` (numeric.rs:225-275, :426-537,
:877-932): Timestamp - Timestamp is a Duration, Timestamp +- Duration keeps the zone, Date32 - Date32 is Duration(Second),
and the library answers every other pair with the reference's InvalidArgumentError text."""
import ctypes as C

from ... import _lib as L
from ... import array as _A
from ...array import Array

ADD, ADD_WRAPPING, SUB, SUB_WRAPPING, MUL, MUL_WRAPPING, DIV, REM = range(8)


def _binary(op, lhs, rhs):
    l, l_s = lhs.get()
    r, r_s = rhs.get()
    ctx = l.ctx
    out = L.ArrayOut()
    lv, rv = l.view(), r.view()
    if l.data_type.logical is None and r.data_type.logical is None:
        ctx.check(ctx.lib.ah_arith_binary(ctx.handle, op, C.byref(lv), int(l_s), C.byref(rv), int(r_s),
                                          C.byref(out)))
        return Array._from_out(ctx, out, l.data_type)
    lt, rt, ot = l.data_type.descriptor(), r.data_type.descriptor(), L.DataTypeDesc()
    ctx.check(ctx.lib.ah_arith_with_types(ctx.handle, op, C.byref(lv), int(l_s), C.byref(lt), C.byref(rv), int(r_s),
                                          C.byref(rt), C.byref(out), C.byref(ot)))
    # Duration + Timestamp swaps inside the library: the zone text comes from whichever side is the Timestamp
    like = l.data_type if l.data_type.logical and l.data_type.logical[0] == ot.id else r.data_type
    return Array._from_out(ctx, out, _A.data_type_from_descriptor(ot, like))


def add(lhs, rhs):
    """numeric.rs:36 — checked for integers, IEEE for floats."""
    return _binary(ADD, lhs, rhs)


def add_wrapping(lhs, rhs):
    """numeric.rs:41"""
    return _binary(ADD_WRAPPING, lhs, rhs)


def sub(lhs, rhs):
    """numeric.rs:46"""
    return _binary(SUB, lhs, rhs)


def sub_wrapping(lhs, rhs):
    """numeric.rs:51"""
    return _binary(SUB_WRAPPING, lhs, rhs)


def mul(lhs, rhs):
    """numeric.rs:56"""
    return _binary(MUL, lhs, rhs)


def mul_wrapping(lhs, rhs):
    """numeric.rs:61"""
    return _binary(MUL_WRAPPING, lhs, rhs)


def div(lhs, rhs):
    """numeric.rs:69"""
    return _binary(DIV, lhs, rhs)


def rem(lhs, rhs):
    """numeric.rs:79"""
    return _binary(REM, lhs, rhs)


def _neg(array, wrapping):
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_arith_neg(ctx.handle, C.byref(v), int(wrapping), C.byref(out)))
    return Array._from_out(ctx, out, array.data_type)


def neg(array):
    """numeric.rs:103"""
    return _neg(array, False)


def neg_wrapping(array):
    """numeric.rs:181-186: `downcast_integer! { … => neg_wrapping, _ => neg(array) }` — only the plain integer DataTypes wrap;
    a Duration / Date / Timestamp column (an integer LAYOUT with a logical type) takes `neg`, which is checked for Duration
    (`neg_wrapping(DurationSecondArray[i64::MIN])` is the overflow error in the reference's test_neg, numeric.rs:1190-1198)."""
    return _neg(array, array.data_type.logical is None)
