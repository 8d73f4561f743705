"""arrow::compute::kernels::sort == arrow_ord::sort (arrow-ord/src/sort.rs): sort_to_indices / sort / sort_limit,
and arrow_ord::partition::partition (arrow-ord/src/partition.rs:126)."""
import ctypes as C
from dataclasses import dataclass

from ... import _lib as L
from ...array import Array, UInt32
from .take import take


@dataclass(frozen=True)
class SortOptions:
    """arrow_schema::SortOptions (arrow-schema/src/lib.rs:87-160); default ASC NULLS FIRST."""
    descending: bool = False
    nulls_first: bool = True

    def __str__(self):
        return ("DESC" if self.descending else "ASC") + (" NULLS FIRST" if self.nulls_first else " NULLS LAST")


def sort_to_indices(values, options=None, limit=None):
    """sort.rs:276 -> UInt32Array of row numbers.  Stable (ties in ascending row order)."""
    options = options or SortOptions()
    ctx = values.ctx
    out = L.ArrayOut()
    v = values.view()
    ctx.check(ctx.lib.ah_sort_to_indices(ctx.handle, C.byref(v), int(options.descending), int(options.nulls_first),
                                         -1 if limit is None else int(limit), C.byref(out)))
    return Array._from_out(ctx, out, UInt32)


def sort(values, options=None):
    """sort.rs:56"""
    return take(values, sort_to_indices(values, options))


def sort_limit(values, options=None, limit=None):
    """sort.rs:158"""
    return take(values, sort_to_indices(values, options, limit))


@dataclass
class SortColumn:
    """sort.rs:870-874"""
    values: object
    options: SortOptions = None


def lexsort_to_indices(columns, limit=None):
    """sort.rs:939: UInt32 row numbers ordering the rows by columns[0], then columns[1], ..."""
    from ...array import InvalidArgumentError
    if not columns:
        raise InvalidArgumentError("Sort requires at least one column")
    ctx = columns[0].values.ctx
    n = len(columns)
    views = (L.ArrayView * n)()
    desc, nf = (C.c_int32 * n)(), (C.c_int32 * n)()
    for i, c in enumerate(columns):
        o = c.options or SortOptions()
        views[i], desc[i], nf[i] = c.values.view(), int(o.descending), int(o.nulls_first)
    out = L.ArrayOut()
    ctx.check(ctx.lib.ah_lexsort_to_indices(ctx.handle, n, views, desc, nf, -1 if limit is None else int(limit),
                                            C.byref(out)))
    return Array._from_out(ctx, out, UInt32)


def lexsort(columns, limit=None):
    """sort.rs:926: every column taken by the lexicographic order."""
    idx = lexsort_to_indices(columns, limit)
    return [take(c.values, idx) for c in columns]


class Partitions:
    """partition.rs:31-80: boundaries between runs of equal rows (of already sorted columns)."""

    def __init__(self, boundaries, num_rows):
        self._bounds, self._n = boundaries, num_rows

    def ranges(self):
        if self._n == 0:
            return []
        cuts = [0] + [b + 1 for b in self._bounds] + [self._n]
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]

    def __len__(self):
        return 0 if self._n == 0 else len(self._bounds) + 1


def partition(columns):
    """partition.rs:126: ranges of consecutive rows equal in every column (`distinct` between neighbours, ORed
    across columns; nulls are equal to nulls).  Runs on the compare / boolean kernels; only the boundary positions
    reach the host."""
    from . import boolean, cmp
    if not columns:
        from ...array import InvalidArgumentError
        raise InvalidArgumentError("Partition requires at least one column")
    n = columns[0].length
    for c in columns:
        if c.length != n:
            from ...array import InvalidArgumentError
            raise InvalidArgumentError("Partition columns have different row counts")
    if n <= 1:
        return Partitions([], n)
    acc = None
    for c in columns:
        d = cmp.distinct(c.slice(0, n - 1), c.slice(1, n - 1))
        acc = d if acc is None else boolean.or_(acc, d)
    # set-bit positions of `acc`: bit i set <=> rows i and i+1 differ
    ctx = acc.ctx
    out = L.ArrayOut()
    v = acc.view()
    # boundaries() of a selection gives flips; the set bits themselves are cheaper through filter(iota, acc)
    from .filter import filter as _filter
    iota = ctx.alloc(max((n - 1) * 4, 8))
    ctx.check(ctx.lib.ah_gen_iota_u32(ctx.handle, iota.ptr, n - 1, 0))
    from ...array import _RawMem
    rows = Array(ctx, UInt32, n - 1, _RawMem(iota.ptr, (n - 1) * 4, iota))
    cuts = _filter(rows, acc)
    return Partitions(cuts.values_numpy().tolist() if cuts.length else [], n)
