"""arrow::compute (arrow/src/compute/mod.rs:22-40): kernels flattened."""
from . import kernels  # noqa: F401
from .kernels.filter import (filter, filter_record_batch, prep_null_mask_filter, FilterBuilder,  # noqa: F401
                             FilterPredicate, filter_expr)
from .kernels.take import take, take_arrays, take_record_batch, TakeOptions  # noqa: F401
from .kernels.numeric import (add, add_wrapping, sub, sub_wrapping, mul, mul_wrapping, div, rem,  # noqa: F401
                              neg, neg_wrapping)
from .kernels.cmp import eq, neq, lt, lt_eq, gt, gt_eq, distinct, not_distinct  # noqa: F401
from .kernels.cast import cast, cast_with_options, cast_chain, can_cast_types, CastOptions  # noqa: F401
from .kernels.concat import concat, concat_batches  # noqa: F401
from .kernels.boolean import (and_, or_, and_not, and_kleene, or_kleene, not_, is_null, is_not_null, nullif)  # noqa: F401
from .kernels.coalesce import BatchCoalescer  # noqa: F401
from .kernels import aggregate  # noqa: F401  (sum/min/max shadow builtins: use ``compute.aggregate.sum`` …)
from .kernels.aggregate import (sum_checked, product, product_checked, bit_and, bit_or, bit_xor,  # noqa: F401
                                min_boolean, max_boolean, bool_and, bool_or)
from .kernels.sort import (sort, sort_limit, sort_to_indices, SortOptions, SortColumn, lexsort, lexsort_to_indices,  # noqa: F401
                           partition, Partitions)
from .kernels.zip import zip  # noqa: F401,A004
from .kernels.interleave import interleave, interleave_record_batch  # noqa: F401
from .kernels.bitwise import (bitwise_and, bitwise_or, bitwise_xor, bitwise_shift_left, bitwise_shift_right,  # noqa: F401
                              bitwise_and_not, bitwise_not, bitwise_and_scalar, bitwise_or_scalar, bitwise_xor_scalar,
                              bitwise_shift_left_scalar, bitwise_shift_right_scalar)
from .kernels.like import like, nlike, starts_with, ends_with, contains, length, bit_length  # noqa: F401,E402
from .kernels.window import shift  # noqa: F401,E402
from .kernels.rank import rank, rank_array  # noqa: F401,E402
