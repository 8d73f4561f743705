"""Child process of tests/test_gpu_deferred.py::test_graph_capture_refuses_calls_that_wait: records calls that cannot be
recorded, checks that each fails fast and that the context keeps working.  TEST INFRASTRUCTURE."""
import faulthandler
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("AH_WORKER_WATCHDOG_S", "200")), exit=True)  # a hang is the failure mode under test: dump and leave

import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402

ctx = A.Context(0)
n = 50_000
rng = np.random.default_rng(5)
a_h, b_h = rng.integers(-1000, 1000, n), rng.integers(-1000, 1000, n)
valid = rng.random(n) < 0.9
a = A.Array.from_numpy(a_h, valid, ctx=ctx)        # nullable: the checked op also unions validity and counts it
b = A.Array.from_numpy(b_h, ctx=ctx)
idx = A.Array.from_numpy(rng.integers(0, n, 100).astype(np.uint32), ctx=ctx)


def works(tag):
    got = K.add_wrapping(a, b)
    assert np.array_equal(got.values_numpy()[valid], (a_h + b_h)[valid]), tag
    assert got.null_count() == int((~valid).sum()), tag
    print("context works", tag, flush=True)


works("before")
cases = {
    "checked add (reads its error word back)": lambda: K.add(a, b),
    "take with check_bounds (reports at return)": lambda: K.take(a, idx, K.TakeOptions(True)),
    "cast to Utf8 (data-dependent size)": lambda: K.cast(b, A.Utf8),
    "a host copy": lambda: a.values_numpy(),
    "synchronize": lambda: ctx.synchronize(),
}
for what, call in cases.items():
    try:
        with ctx.graph_capture():
            call()
        raise AssertionError(f"{what}: recording must fail")
    except (A.ArrowError, A.array.HipError) as ex:
        print(f"refused as it should be: {what}: {str(ex)[:120]}", flush=True)
    works(f"after '{what}'")
# a good capture still works afterwards
with ctx.graph_capture() as g:
    out = K.add_wrapping(a, b)
g.launch()
ctx.synchronize()
assert np.array_equal(out.values_numpy()[valid], (a_h + b_h)[valid])
print("GRAPH_FAIL_WORKER_OK", flush=True)
