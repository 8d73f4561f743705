"""arrow::compute::kernels::bitwise == arrow_arith::bitwise (arrow-arith/src/bitwise.rs).  The ``_scalar`` forms take a
``Scalar`` on the right, like the reference takes a native value."""
import ctypes as C

from ... import _lib as L
from ...array import Array
from .numeric import _binary

AND, OR, XOR, SHIFT_LEFT, SHIFT_RIGHT, AND_NOT = range(8, 14)


def bitwise_and(left, right):
    """bitwise.rs:42"""
    return _binary(AND, left, right)


def bitwise_or(left, right):
    """bitwise.rs:55"""
    return _binary(OR, left, right)


def bitwise_xor(left, right):
    """bitwise.rs:68"""
    return _binary(XOR, left, right)


def bitwise_shift_left(left, right):
    """bitwise.rs:81 — ``wrapping_shl``: the count is taken modulo the bit width"""
    return _binary(SHIFT_LEFT, left, right)


def bitwise_shift_right(left, right):
    """bitwise.rs:97 — ``wrapping_shr``: arithmetic for signed types"""
    return _binary(SHIFT_RIGHT, left, right)


def bitwise_and_not(left, right):
    """bitwise.rs:123"""
    return _binary(AND_NOT, left, right)


def bitwise_not(array):
    """bitwise.rs:113"""
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_bitwise_not(ctx.handle, C.byref(v), C.byref(out)))
    return Array._from_out(ctx, out, array.data_type)


bitwise_and_scalar, bitwise_or_scalar, bitwise_xor_scalar = bitwise_and, bitwise_or, bitwise_xor  # :137-175
bitwise_shift_left_scalar, bitwise_shift_right_scalar = bitwise_shift_left, bitwise_shift_right  # :176-205
