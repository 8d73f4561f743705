"""arrow::compute::kernels::{like, length} == arrow_string::like / arrow_string::length
(arrow-string/src/like.rs:83-205, length.rs:58-140) for Utf8 / LargeUtf8 columns and scalar patterns."""
import ctypes as C

from ... import _lib as L
from ...array import Array, Boolean, Int32, Int64, Scalar

LIKE, NLIKE, STARTS_WITH, ENDS_WITH, CONTAINS = range(5)


def _apply(op, left, right):
    arr, l_s = left.get()
    pat, r_s = right.get()
    if l_s:
        raise NotImplementedError("string predicates with a scalar on the left")
    ctx = arr.ctx
    out = L.ArrayOut()
    lv, rv = arr.view(), pat.view()
    ctx.check(ctx.lib.ah_string_like(ctx.handle, op, C.byref(lv), C.byref(rv), 1 if r_s else 0, C.byref(out)))
    return Array._from_out(ctx, out, Boolean)


def _pattern(left, pattern):
    """Accept a plain ``str`` for the pattern, like ``StringArray::new_scalar(..)`` on the reference side."""
    if isinstance(pattern, str):
        arr, _ = left.get()
        return Scalar(Array.from_strings([pattern], None, arr.data_type, arr.ctx))
    return pattern


def like(left, right):
    """like.rs:83 — SQL ``left LIKE right``: ``%`` any run of characters, ``_`` one character, ``\\`` escapes."""
    return _apply(LIKE, left, _pattern(left, right))


def nlike(left, right):
    """like.rs:103 — ``left NOT LIKE right``"""
    return _apply(NLIKE, left, _pattern(left, right))


def starts_with(left, right):
    """like.rs:138"""
    return _apply(STARTS_WITH, left, _pattern(left, right))


def ends_with(left, right):
    """like.rs:164"""
    return _apply(ENDS_WITH, left, _pattern(left, right))


def contains(left, right):
    """like.rs:190"""
    return _apply(CONTAINS, left, _pattern(left, right))


def _length(array, bits):
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_string_length(ctx.handle, C.byref(v), bits, C.byref(out)))
    return Array._from_out(ctx, out, Int32 if array.data_type.physical == L.AH_UTF8 else Int64)


def length(array):
    """length.rs:58 — byte length of every string (Int32 for Utf8, Int64 for LargeUtf8)."""
    return _length(array, 0)


def bit_length(array):
    """length.rs:130"""
    return _length(array, 1)
