"""Helpers on Boolean arrays used by the filter path (arrow-arith/src/boolean.rs is a
"next" row; only what prep_null_mask_filter needs lives here for now)."""
from ...array import Array, Boolean, Scalar


def _and_validity(filter_array):
    """values AND validity as a new BooleanArray without nulls (filter.rs:167-171),
    expressed with the device compare kernel: not_distinct(x, true) is true exactly
    where x is valid and true."""
    from .cmp import not_distinct
    if filter_array.validity is None:
        return filter_array
    return not_distinct(filter_array, Scalar.new(True, Boolean, filter_array.ctx))
