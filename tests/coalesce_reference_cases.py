"""The deterministic scenario tests of arrow-select/src/coalesce.rs, transcribed (TEST INFRASTRUCTURE): every test of the
reference's `mod tests` whose input and expected output sizes are fixed numbers (the filtered ones draw their masks from a Rust
RNG and are covered by the model-based fuzz instead).  A scenario is a list of steps run against an adapter — the device
coalescer (tests/test_gpu_parity.py) or the Python model the fuzz tests compare with (tests/test_oracle_golden.py, CPU):
    ("push", n)              push a batch of n rows, column c0 = 0 .. n-1         (create_test_batch / uint32_batch)
    ("drain", [sizes])       next_completed_batch() until None: exactly these row counts, in order
    ("buffered", k)          get_buffered_rows() == k
    ("has_completed", b)     has_completed_batch() == b
    ("finish",)              finish_buffered_batch()
Every scenario ends with an implicit finish + drain of `tail`; the concatenation of ALL output rows must equal the concatenation
of all pushed rows (what the reference's `Test::run` checks against concat_batches, coalesce.rs:1960-2010)."""

S = "arrow-select/src/coalesce.rs"

SCENARIOS = [
    # name, source, target, biggest_coalesce_batch_size, steps, tail (sizes drained after the final finish)
    ("test_coalesce", f"{S}:794-802", 21, None, [("push", 8)] * 10, [21, 21, 21, 17]),
    ("test_coalesce_one_by_one", f"{S}:805-813", 20, None, [("push", 1)] * 97, [20, 20, 20, 20, 17]),
    ("test_coalesce_empty", f"{S}:816-825", 21, None, [], []),
    ("test_single_large_batch_greater_than_target", f"{S}:835-843", 1000, None, [("push", 4096)], [1000, 1000, 1000, 1000, 96]),
    ("test_single_large_batch_smaller_than_target", f"{S}:846-854", 8192, None, [("push", 4096)], [4096]),
    ("test_single_large_batch_equal_to_target", f"{S}:857-865", 4096, None, [("push", 4096)], [4096]),
    ("test_single_large_batch_equally_divisible_in_target", f"{S}:868-876", 1024, None, [("push", 4096)], [1024] * 4),
    ("test_coalesce_non_null", f"{S}:1024-1032", 1024, None, [("push", 3000), ("push", 1040)], [1024, 1024, 1024, 968]),
    ("test_biggest_coalesce_batch_size_none_default", f"{S}:2210-2237", 100, None, [("push", 1000)], [100] * 10),
    ("test_biggest_coalesce_batch_size_bypass_large_batch", f"{S}:2240-2260", 100, 500,
     [("push", 1000), ("has_completed", True), ("drain", [1000]), ("has_completed", False), ("buffered", 0)], []),
    ("test_biggest_coalesce_batch_size_coalesce_small_batch", f"{S}:2263-2293", 100, 500,
     [("push", 50), ("has_completed", False), ("buffered", 50), ("push", 50), ("has_completed", True), ("drain", [100]), ("buffered", 0)], []),
    ("test_biggest_coalesce_batch_size_equal_boundary", f"{S}:2296-2321", 100, 500, [("push", 500)], [100] * 5),
    ("test_biggest_coalesce_batch_size_first_large_then_consecutive_bypass", f"{S}:2324-2369", 100, 200,
     [("push", 50), ("buffered", 50), ("has_completed", False), ("push", 250), ("drain", [100, 100, 100]), ("buffered", 0),
      ("push", 300), ("has_completed", True), ("drain", [300]), ("buffered", 0), ("push", 400), ("has_completed", True), ("drain", [400]),
      ("buffered", 0)], []),
    ("test_biggest_coalesce_batch_size_empty_batch", f"{S}:2372-2386", 100, 50, [("push", 0), ("has_completed", False), ("buffered", 0)], []),
    ("test_biggest_coalesce_batch_size_with_buffered_data_no_bypass", f"{S}:2389-2421", 100, 200,
     [("push", 30), ("push", 30), ("buffered", 60), ("push", 250), ("drain", [100, 100, 100]), ("buffered", 10)], [10]),
    ("test_biggest_coalesce_batch_size_zero_limit", f"{S}:2424-2439", 100, 0, [("push", 1), ("has_completed", True), ("drain", [1])], []),
    ("test_biggest_coalesce_batch_size_bypass_only_when_no_buffer", f"{S}:2442-2479", 100, 200,
     [("push", 300), ("has_completed", True), ("drain", [300]), ("buffered", 0), ("push", 50), ("buffered", 50), ("push", 300),
      ("drain", [100, 100, 100]), ("buffered", 50)], [50]),
    ("test_biggest_coalesce_batch_size_consecutive_large_batches_scenario", f"{S}:2482-2528", 1000, 500,
     [("push", 20), ("push", 20), ("push", 30), ("buffered", 70), ("has_completed", False), ("push", 700), ("buffered", 770),
      ("has_completed", False), ("push", 600), ("drain", [770, 600]), ("buffered", 0)]
     + [step for size in (700, 900, 700, 600) for step in (("push", size), ("has_completed", True), ("drain", [size]), ("buffered", 0))], []),
    ("test_biggest_coalesce_batch_size_truly_consecutive_large_bypass", f"{S}:2531-2594", 100, 200,
     [step for size in (300, 400, 350, 500)
      for step in (("buffered", 0), ("push", size), ("has_completed", True), ("drain", [size]), ("has_completed", False), ("buffered", 0))], []),
    ("test_biggest_coalesce_batch_size_reset_consecutive_on_small_batch", f"{S}:2597-2632", 100, 200,
     [("push", 300), ("drain", [300]), ("push", 400), ("drain", [400]), ("push", 50), ("buffered", 50), ("push", 350),
      ("drain", [100, 100, 100, 100]), ("buffered", 0)], []),
]


def run_scenario(adapter, steps, tail):
    """adapter: push(n) / drain() -> [list of rows as python lists] / buffered() / has_completed() / finish().
    Returns nothing; raises AssertionError with the step index."""
    pushed, got = [], []

    def drain(expect, where):
        batches = adapter.drain()
        assert [len(b) for b in batches] == list(expect), f"{where}: output sizes {[len(b) for b in batches]} != {list(expect)}"
        for b in batches:
            got.extend(b)

    for i, st in enumerate(steps):
        where = f"step {i} {st}"
        if st[0] == "push":
            adapter.push(st[1])
            pushed.extend(range(st[1]))
        elif st[0] == "drain":
            drain(st[1], where)
        elif st[0] == "buffered":
            assert adapter.buffered() == st[1], f"{where}: buffered {adapter.buffered()}"
        elif st[0] == "has_completed":
            assert adapter.has_completed() == st[1], where
        elif st[0] == "finish":
            adapter.finish()
        else:
            raise ValueError(st)
    adapter.finish()
    drain(tail, "after the final finish")
    assert got == pushed, "the output rows are not the pushed rows in order"
