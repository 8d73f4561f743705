import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# Soak runs (tools/soak_gpu.sh): AH_SEED_OFFSET=k shifts every np.random.default_rng(seed) of the suite, so the same
# randomized parity tests draw different inputs.  Unset (the driver's runs): the fixed seeds, reproducible.
_SEED_OFFSET = int(os.environ.get("AH_SEED_OFFSET", "0") or 0)
if _SEED_OFFSET:
    import numpy as _np
    _orig_default_rng = _np.random.default_rng

    def _shifted_default_rng(seed=None, *a, **k):
        if isinstance(seed, int):
            seed = seed + 1_000_003 * _SEED_OFFSET
        return _orig_default_rng(seed, *a, **k)

    _np.random.default_rng = _shifted_default_rng


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_logstart(nodeid, location):
    """The id of the test about to run, straight to fd 2 (unbuffered, no capture): when the GPU faults the HSA runtime
    aborts the process at once, and the last `[ah-test]` line of the log is then the test that was running."""
    try:
        os.write(2, f"\n[ah-test] {nodeid}\n".encode())
    except OSError:
        pass


def pytest_sessionstart(session):
    """A clean checkout has no built artefacts (*.so is git-ignored): build the HIP library (hipcc cross-compiles
    without a GPU) before collection imports the package.  The product itself never builds or falls back."""
    lib = os.path.join(ROOT, "arrow-rs_amd", "lib", "libarrow_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "arrow-rs_amd", "csrc"), "-j", str(min(16, os.cpu_count() or 4))])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so) — test infrastructure only."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import orc
    return orc.load(so)


@pytest.fixture(scope="session")
def ctx():
    """A device context. Only gpu-marked tests may request it: it raises without a GPU."""
    import arrow_rs_amd as A
    import numpy as np
    os.write(2, b"\n[ah-test] session: creating the context\n")
    c = A.Context(0)
    A.set_default_context(c)
    # runtime copies only, before the first kernel of the library runs: a fault between these two lines is the box's, not a kernel's
    os.write(2, b"[ah-test] session: HIP self test (1 MiB host -> device -> host, runtime copies only)\n")
    src = (np.arange(1 << 20, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(24)).astype(np.uint8)
    assert np.array_equal(A.array.DeviceBuffer.from_numpy(c, src).to_numpy(), src), "HIP self test: the round trip changed bytes"
    os.write(2, b"[ah-test] session: HIP self test ok\n")
    return c
