"""arrow::compute::kernels (arrow/src/compute/kernels.rs:20-27)."""
from . import filter, take, numeric, cmp, cast, concat, boolean, coalesce, aggregate, sort, zip, interleave, bitwise  # noqa: F401
