"""arrow::compute::kernels::boolean == arrow_arith::boolean (arrow-arith/src/boolean.rs).
``and`` / ``or`` / ``not`` are Python keywords, so those three carry a trailing underscore."""
import ctypes as C

from ... import _lib as L
from ...array import Array, Boolean

AND, OR, AND_NOT, AND_KLEENE, OR_KLEENE = 0, 1, 2, 3, 4
NOT, IS_NULL, IS_NOT_NULL = 10, 11, 12


def _binary(op, left, right):
    ctx = left.ctx
    out = L.ArrayOut()
    lv, rv = left.view(), right.view()
    ctx.check(ctx.lib.ah_boolean_binary(ctx.handle, op, C.byref(lv), C.byref(rv), C.byref(out)))
    return Array._from_out(ctx, out, Boolean)


def _unary(op, values):
    ctx = values.ctx
    out = L.ArrayOut()
    v = values.view()
    ctx.check(ctx.lib.ah_boolean_unary(ctx.handle, op, C.byref(v), C.byref(out)))
    return Array._from_out(ctx, out, Boolean)


def and_(left, right):
    """boolean.rs:256"""
    return _binary(AND, left, right)


def or_(left, right):
    """boolean.rs:273"""
    return _binary(OR, left, right)


def and_not(left, right):
    """boolean.rs:291"""
    return _binary(AND_NOT, left, right)


def and_kleene(left, right):
    """boolean.rs:60"""
    return _binary(AND_KLEENE, left, right)


def or_kleene(left, right):
    """boolean.rs:156"""
    return _binary(OR_KLEENE, left, right)


def not_(values):
    """boolean.rs:310"""
    return _unary(NOT, values)


def is_null(values):
    """boolean.rs:327"""
    return _unary(IS_NULL, values)


def is_not_null(values):
    """boolean.rs:347"""
    return _unary(IS_NOT_NULL, values)


def _and_validity(filter_array):
    """prep_null_mask_filter (arrow-select/src/filter.rs:167-171): values AND validity, no nulls."""
    if filter_array.validity is None:
        return filter_array
    return and_(Array(filter_array.ctx, Boolean, filter_array.length, filter_array.values,
                      filter_array.values_bit_offset),
                is_not_null(filter_array))


def nullif(left, right):
    """``arrow_select::nullif::nullif`` (arrow-select/src/nullif.rs:60): zero-copy values, new nulls."""
    ctx = left.ctx
    out = L.ArrayOut()
    lv, rv = left.view(), right.view()
    ctx.check(ctx.lib.ah_nullif(ctx.handle, C.byref(lv), C.byref(rv), C.byref(out)))
    return Array._from_out(ctx, out, left.data_type, keepalive=(left,))
