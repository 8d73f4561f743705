"""Host-side hardening of the C ABI (the reference runs Miri over the crates that parse
untrusted bytes, .github/workflows/miri.sh:12-45).

`make -C arrow-rs_amd/csrc SAN=1` builds the HOST half of the whole library with AddressSanitizer + UBSan (clang ignores
-fsanitize for gfx950 device code).  On this CPU-only box the sanitized library then
  * loads, and exports the same symbols as the product build,
  * takes 2 x 60 000 mutated IPC messages / file footers through the host-only decoders (tests/cpp/ipc_fuzz_host.cpp:
    byte flips, boundary values in offset / length words, truncations, splices, noise): every outcome must be a status,
    never a sanitizer report.  (Round 4's first run of this fuzz found one: a negative FloatingPoint precision indexed
    "efg"[-2] in csrc/ipc.hip — fixed, and pinned below.)
The first build takes ~1.5 min (31 translation units), later runs are incremental; AH_SKIP_SAN=1 skips the file."""
import ctypes
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "arrow-rs_amd", "csrc")
SAN_DIR = os.environ.get("AH_SAN_DIR", "/tmp/arrow_hip_san")  # outside the tree: 110 MB of instrumented objects
SAN_LIB = os.path.join(SAN_DIR, "libarrow_hip_san.so")
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")

pytestmark = pytest.mark.skipif(os.environ.get("AH_SKIP_SAN") == "1", reason="AH_SKIP_SAN=1")


@pytest.fixture(scope="module")
def san_lib():
    subprocess.check_call(["make", "-C", CSRC, "SAN=1", f"SANDIR={SAN_DIR}", "-j", str(min(16, os.cpu_count() or 4))], stdout=subprocess.DEVNULL)
    assert os.path.exists(SAN_LIB)
    return SAN_LIB


@pytest.fixture(scope="module")
def fuzz_exe(san_lib, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "ipc_fuzz_host")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-Wno-option-ignored", os.path.join(ROOT, "tests", "cpp", "ipc_fuzz_host.cpp"), "-L" + os.path.dirname(san_lib),
                           "-larrow_hip_san", "-Wl,-rpath," + os.path.dirname(san_lib), "-o", exe])
    return exe


def test_sanitized_library_exports_the_abi(san_lib):
    """Same symbols as the product build (the sanitizer build is the same sources with other flags): checked in a child
    process, because an ASan runtime cannot be loaded late into this interpreter."""
    import re
    hdr = open(os.path.join(ROOT, "include", "arrow_hip.h")).read()
    syms = sorted(set(re.findall(r"AH_API\s+[\w\s\*]+?\b(ah_\w+)\s*\(", hdr)))
    out = subprocess.run(["nm", "-D", "--defined-only", san_lib], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    assert "__asan_init" in subprocess.run(["nm", "-D", san_lib], capture_output=True, text=True).stdout  # really instrumented


@pytest.mark.parametrize("seed", [1, 2])
def test_ipc_decoders_survive_mutated_messages(fuzz_exe, seed):
    r = subprocess.run([fuzz_exe, "60000", str(seed)], capture_output=True, text=True, timeout=600, env=SAN_ENV)
    assert r.returncode == 0 and "IPC_FUZZ_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    # the mutations must reach the decoders' error paths AND leave some messages acceptable (else nothing was parsed)
    tally = dict(tuple(l.strip().rsplit(": ", 1)) for l in r.stdout.splitlines() if l.startswith("  "))
    assert int(tally["decode_schema status 0"]) > 1000 and int(tally["decode_footer status 0"]) > 1000
    errors = sum(int(v) for k, v in tally.items() if not k.endswith("status 0"))
    assert errors > 50_000, tally


def test_negative_float_precision_is_an_error_not_an_index():
    """The bug the fuzz found, as a directed case through the PRODUCT library: a Schema message whose FloatingPoint
    precision word is -2 must come back as AH_IPC_ERROR (it read "efg"[-2] before)."""
    import arrow_rs_amd as A
    lib = A._lib.load()
    F = A._lib.IpcField if hasattr(A._lib, "IpcField") else None
    if F is None:
        class F(ctypes.Structure):
            _fields_ = [("name", ctypes.c_char_p), ("format", ctypes.c_char_p), ("nullable", ctypes.c_int32)]
    fields = (F * 1)(F(b"x", b"g", 1))
    out, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_int64()
    lib.ah_ipc_schema_message.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_int64)]
    assert lib.ah_ipc_schema_message(None, 1, fields, 8, ctypes.byref(out), ctypes.byref(n)) == 0
    msg = bytearray(ctypes.string_at(out, n.value))
    lib.ah_host_free.argtypes = [ctypes.c_void_p]
    lib.ah_host_free(out)
    # the FloatingPoint table holds one i16: DOUBLE = 2.  Find the 16-bit words equal to 2 and try each as "the precision":
    # at least one of them must turn the message into an AH_IPC_ERROR about the precision, none may crash
    lib.ah_ipc_decode_schema.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32),
                                         ctypes.POINTER(ctypes.c_void_p)]
    hits = 0
    for at in range(8, len(msg) - 1, 2):
        if struct.unpack_from("<h", msg, at)[0] != 2:
            continue
        m = bytearray(msg)
        struct.pack_into("<h", m, at, -2)
        nf, fl = ctypes.c_int32(), ctypes.c_void_p()
        st = lib.ah_ipc_decode_schema(None, bytes(m), len(m), ctypes.byref(nf), ctypes.byref(fl))
        if st == 0:
            lib.ah_host_free(fl)
        elif st == A._lib.AH_IPC_ERROR:
            hits += 1
    assert hits >= 1
