"""PCIe-inclusive rate of the C Data boundary: host pyarrow array -> ah_import_c_data
(H2D) -> filter -> ah_export_c_data (D2H).  Never the bench `value` (that starts HBM-resident)."""
import json
import os
import sys
import time

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
ctx = A.Context(0)
rng = np.random.default_rng(1)
vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.1)
mask = pa.array(rng.random(n) < 0.1)
res = {"rows": n, "value_bytes": n * 8}
for rep in range(3):
    t0 = time.perf_counter()
    dv = A.Array.from_pyarrow(vals, ctx)
    dm = A.Array.from_pyarrow(mask, ctx)
    t1 = time.perf_counter()
    f = K.filter(dv, dm)
    ctx.synchronize()
    t2 = time.perf_counter()
    back = f.to_pyarrow()
    t3 = time.perf_counter()
    res = dict(res, import_s=round(t1 - t0, 4), filter_s=round(t2 - t1, 5), export_s=round(t3 - t2, 4),
               import_GBps=round((n * 8 + n / 4) / (t1 - t0) / 1e9, 2),
               export_GBps=round(len(back) * 8.125 / (t3 - t2) / 1e9, 2),
               end_to_end_Mrows_per_s=round(n / (t3 - t0) / 1e6, 1), selected=len(back))
print(json.dumps(res))
