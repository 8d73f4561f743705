"""arrow::compute::kernels::numeric == arrow_arith::numeric (arrow-arith/src/numeric.rs:36-186).
All binary kernels take two ``Datum`` (``Array`` or ``Scalar``)."""
import ctypes as C

from ... import _lib as L
from ...array import Array

ADD, ADD_WRAPPING, SUB, SUB_WRAPPING, MUL, MUL_WRAPPING, DIV, REM = range(8)


def _binary(op, lhs, rhs):
    l, l_s = lhs.get()
    r, r_s = rhs.get()
    ctx = l.ctx
    out = L.ArrayOut()
    lv, rv = l.view(), r.view()
    ctx.check(ctx.lib.ah_arith_binary(ctx.handle, op, C.byref(lv), int(l_s), C.byref(rv), int(r_s),
                                      C.byref(out)))
    return Array._from_out(ctx, out, l.data_type)


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
    """numeric.rs:181"""
    return _neg(array, True)
