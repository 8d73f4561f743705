"""csrc/big_int_digits.hpp on the CPU (no GPU needed): the PRODUCT header cast_string.hip compiles for gfx950 gives integer-valued
doubles in [2^53, 2^64) — the general rows of the Int64 -> Float64 -> Utf8 chain (arrow-cast/src/cast/string.rs:21-39 ->
display.rs:711-723, `ryu::Buffer::format`) — their shortest round-trip digits from Ryu's digit-removal loop on the exact 64-bit
interval.  tests/cpp/big_int_digits_host_test.cpp compares it with libstdc++'s shortest std::to_chars (itself Ryu) on random
mantissas at every exponent, the neighbours of 10^k / 5 * 10^(k-1) (ties, long removable runs), multiples of 10^j and the class
boundaries (powers of two and values outside must be declined).  The GPU side of the same function:
tests/test_gpu_parity.py::test_cast_big_integer_doubles_to_utf8."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_big_int_digits_header_matches_shortest_to_chars(tmp_path):
    exe = str(tmp_path / "big_int_digits_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "big_int_digits_host_test.cpp")],
                   check=True)
    total = 0
    for seed in ("1", "2", "3"):
        r = subprocess.run([exe, "1000000", seed], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "BIG_INT_DIGITS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        total += int(r.stdout.split()[-1])
    assert total > 3 * 11_000_000
