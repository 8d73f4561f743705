"""The C-ABI library loads on a CPU-only box and exports every symbol include/arrow_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "arrow_hip.h")).read()
    return sorted(set(re.findall(r"AH_API\s+[\w\s\*]+?\b(ah_\w+)\s*\(", hdr)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert len(syms) >= 35
    for must in ["ah_filter", "ah_take", "ah_arith_binary", "ah_compare", "ah_cast", "ah_concat"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    import arrow_rs_amd as A
    lib = ctypes.CDLL(A._lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libarrow_hip.so does not export: {missing}"


def test_binding_covers_header():
    import arrow_rs_amd as A
    assert sorted(A._lib.SIGNATURES) == declared_symbols()
    lib = A._lib.load()
    assert lib.ah_version().startswith(b"arrow_hip")


def test_rust_sys_bindings_are_current():
    """bindings/rust/arrow-hip-sys/src/lib.rs is generated from the header (tools/gen_rust_sys.py): it must be up to
    date, declare every exported symbol, mirror the struct layouts field for field, and every `sys::` name the
    hand-written safe crate uses must exist in it.  (No Rust toolchain here: this is the only check it gets.)"""
    import importlib.util
    import arrow_rs_amd as A
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    assert committed == gen.generate(), "run `python tools/gen_rust_sys.py`"
    declared = re.findall(r"pub fn (ah_\w+)\(", committed)
    assert sorted(declared) == declared_symbols()
    # struct layouts: same field names, in order, as the ctypes mirror the GPU tests run through
    for rust_name, ct in [("ah_array_view", A._lib.ArrayView), ("ah_array_out", A._lib.ArrayOut)]:
        body = re.search(r"pub struct %s \{(.*?)\}" % rust_name, committed, flags=re.S).group(1)
        fields = [f.rstrip("_") for f in re.findall(r"pub (\w+):", body)]
        assert fields == [f[0] for f in ct._fields_], rust_name
    safe = open(os.path.join(ROOT, "bindings", "rust", "arrow-hip", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(\w+)", safe))
    known = set(re.findall(r"pub (?:fn|const|struct|type) (\w+)", committed))
    assert used <= known, f"safe crate uses undeclared names: {sorted(used - known)}"


def test_no_cpu_fallback_without_gpu():
    """Product path must fail loudly when no HIP device is usable."""
    import arrow_rs_amd as A
    lib = A._lib.load()
    h = ctypes.c_void_p()
    st = lib.ah_context_create(0, ctypes.byref(h))
    if st == 0:  # a GPU is present (GPU box): nothing to assert here
        lib.ah_context_destroy(h)
        pytest.skip("GPU present")
    with pytest.raises(A.HipError):
        A.Context(0)


def test_product_does_not_reference_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "arrow-rs_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) in ("build", "lib"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"liboracle|oracle/|orc_\w+\(|import orc\b", txt):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders


def test_header_is_valid_c99(tmp_path):
    """include/arrow_hip.h is a C header (the FFI boundary): it must compile as plain C and link
    against the shared library without any C++ or HIP header."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "arrow_hip.h"\n'
                   '#include <stdio.h>\n'
                   'int main(void) {\n'
                   '  ah_array_view v = {0}; ah_array_out o = {0};\n'
                   '  v.type = AH_INT64; (void)o;\n'
                   '  printf("%s %d %d\\n", ah_version(), (int)sizeof(ah_array_view), ah_can_cast_types(AH_INT64, AH_FLOAT64));\n'
                   '  return v.type == AH_INT64 ? 0 : 1;\n'
                   '}\n')
    import arrow_rs_amd as A
    libdir = os.path.dirname(A._lib.LIB_PATH)
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src),
                           "-L" + libdir, "-larrow_hip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.startswith("arrow_hip") and out.stdout.strip().endswith(" 1"), out


def test_cpp_host_mirror_compiles():
    """include/arrow_hip.hpp + the C++ mirror test compile with plain g++ (no HIP headers)."""
    import subprocess
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")])


def test_c_data_format_mapping():
    """ah_type_from_format: DataType::try_from(&FFI_ArrowSchema) reduced to physical layouts
    (arrow-schema/src/ffi.rs:492-700).  Pure host logic: needs no GPU and no context."""
    import ctypes as C
    import arrow_rs_amd as A
    L = A._lib
    lib = L.load()
    cases = {
        "b": L.AH_BOOL, "c": L.AH_INT8, "C": L.AH_UINT8, "s": L.AH_INT16, "S": L.AH_UINT16, "i": L.AH_INT32,
        "I": L.AH_UINT32, "l": L.AH_INT64, "L": L.AH_UINT64, "e": L.AH_FLOAT16, "f": L.AH_FLOAT32,
        "g": L.AH_FLOAT64, "u": L.AH_UTF8, "U": L.AH_LARGE_UTF8, "z": L.AH_UTF8, "Z": L.AH_LARGE_UTF8,
        "tdD": L.AH_INT32, "tdm": L.AH_INT64, "tts": L.AH_INT32, "ttm": L.AH_INT32, "ttu": L.AH_INT64,
        "ttn": L.AH_INT64, "tDs": L.AH_INT64, "tDn": L.AH_INT64, "tiM": L.AH_INT32, "tiD": L.AH_INT64,
        "tin": L.AH_FIXED16, "tss:": L.AH_INT64, "tsu:UTC": L.AH_INT64, "tsn:America/New_York": L.AH_INT64,
        "d:38,10": L.AH_FIXED16, "d:9,2,32": L.AH_INT32, "d:18,2,64": L.AH_INT64, "d:38,10,128": L.AH_FIXED16,
        "d:76,10,256": L.AH_FIXED32, "w:16": L.AH_FIXED16, "w:32": L.AH_FIXED32,
    }
    for fmt, want in cases.items():
        t = C.c_int32(0)
        assert lib.ah_type_from_format(None, fmt.encode(), C.byref(t)) == L.AH_OK, fmt
        assert t.value == want, fmt
    t = C.c_int32(0)
    for fmt in ("+l", "+s", "vu", "vz", "n", "w:3"):
        assert lib.ah_type_from_format(None, fmt.encode(), C.byref(t)) == L.AH_NOT_YET_IMPLEMENTED, fmt
    for fmt in ("x", "d:1", "d:a,b", "d:10,2,512", "tsx:", ""):
        assert lib.ah_type_from_format(None, fmt.encode(), C.byref(t)) == L.AH_C_DATA_INTERFACE, fmt
    for phys in range(1, 17):
        fmt = lib.ah_format_of_type(phys)
        assert fmt is not None
        assert lib.ah_type_from_format(None, fmt, C.byref(t)) == L.AH_OK and t.value == phys


def _ipc_schema_cases():
    import pyarrow as pa
    return pa.schema([
        pa.field("i8", pa.int8()), pa.field("u16", pa.uint16(), nullable=False), pa.field("i32", pa.int32()),
        pa.field("u64", pa.uint64()), pa.field("f16", pa.float16()), pa.field("f32", pa.float32()),
        pa.field("f64", pa.float64(), nullable=False), pa.field("b", pa.bool_()), pa.field("s", pa.string()),
        pa.field("ls", pa.large_string()), pa.field("bin", pa.binary()), pa.field("lbin", pa.large_binary()),
        pa.field("d32", pa.date32()), pa.field("d64", pa.date64()), pa.field("t32s", pa.time32("s")),
        pa.field("t32ms", pa.time32("ms")), pa.field("t64us", pa.time64("us")), pa.field("t64ns", pa.time64("ns")),
        pa.field("ts_s", pa.timestamp("s")), pa.field("ts_us_utc", pa.timestamp("us", tz="UTC")),
        pa.field("ts_ns_ny", pa.timestamp("ns", tz="America/New_York")), pa.field("dur_ms", pa.duration("ms")),
        pa.field("dur_ns", pa.duration("ns")), pa.field("dec128", pa.decimal128(20, 3)),
        pa.field("dec256", pa.decimal256(50, 7)), pa.field("mdn", pa.month_day_nano_interval()),
        pa.field("fsb16", pa.binary(16)), pa.field("", pa.int64()), pa.field("ünïcode ✓", pa.int64()),
    ])


def test_ipc_schema_message_round_trips_through_pyarrow():
    """Host-only half of the IPC boundary: our hand-written flatbuffers against Arrow C++'s verifier/reader, and
    Arrow C++'s writer against our reader (arrow-ipc/src/convert.rs `schema_to_fb` / `fb_to_schema`)."""
    import ctypes as C
    import pyarrow as pa
    import arrow_rs_amd as A
    L = A._lib
    lib = L.load()
    sch = _ipc_schema_cases()
    # pyarrow -> ours
    msg = sch.serialize().to_pybytes()
    n, fields = C.c_int32(), C.POINTER(L.IpcField)()
    assert lib.ah_ipc_decode_schema(None, msg, len(msg), C.byref(n), C.byref(fields)) == L.AH_OK
    got = [(fields[i].name.decode(), fields[i].format.decode(), bool(fields[i].nullable)) for i in range(n.value)]
    lib.ah_host_free(fields)
    want_fmt = ["c", "S", "i", "L", "e", "f", "g", "b", "u", "U", "z", "Z", "tdD", "tdm", "tts", "ttm", "ttu", "ttn",
                "tss:", "tsu:UTC", "tsn:America/New_York", "tDm", "tDn", "d:20,3", "d:50,7,256", "tin", "w:16", "l", "l"]
    assert got == [(f.name, fmt, f.nullable) for f, fmt in zip(sch, want_fmt)]
    # ours -> pyarrow, at every legal alignment
    for alignment in (8, 16, 32, 64):
        arr = (L.IpcField * len(got))()
        keep = [(nm.encode(), fm.encode()) for nm, fm, _ in got]
        for i, (nm, fm) in enumerate(keep):
            arr[i].name, arr[i].format, arr[i].nullable = nm, fm, 1 if got[i][2] else 0
        out, ln = C.c_void_p(), C.c_int64()
        assert lib.ah_ipc_schema_message(None, len(got), arr, alignment, C.byref(out), C.byref(ln)) == L.AH_OK
        blob = C.string_at(out, ln.value)
        lib.ah_host_free(out)
        assert ln.value % alignment == 0 and blob[:4] == b"\xff" * 4
        assert int.from_bytes(blob[4:8], "little") == ln.value - 8
        back = pa.ipc.read_schema(pa.py_buffer(blob))
        assert back.equals(sch), back
        ht, bl = C.c_int32(), C.c_int64()
        assert lib.ah_ipc_message_info(None, blob, len(blob), C.byref(ht), C.byref(bl)) == L.AH_OK
        assert (ht.value, bl.value) == (1, 0)
    out, ln = C.c_void_p(), C.c_int64()
    assert lib.ah_ipc_schema_message(None, 0, None, 12, C.byref(out), C.byref(ln)) == L.AH_INVALID_ARGUMENT
    # nested / dictionary schemas are refused, truncated metadata is a parse/ipc error, never a crash
    for bad in (pa.schema([pa.field("l", pa.list_(pa.int32()))]), pa.schema([pa.field("d", pa.dictionary(pa.int8(), pa.string()))])):
        m = bad.serialize().to_pybytes()
        assert lib.ah_ipc_decode_schema(None, m, len(m), C.byref(n), C.byref(fields)) == L.AH_NOT_YET_IMPLEMENTED
    for cut in (3, 8, 20, len(msg) // 2):
        st = lib.ah_ipc_decode_schema(None, msg[:cut], cut, C.byref(n), C.byref(fields))
        assert st in (L.AH_IPC_ERROR, L.AH_PARSE_ERROR), (cut, st)


def test_ipc_file_footer_against_pyarrow():
    """Host-only half of the IPC FILE format (FileWriter::finish writer.rs:1724, read_footer_length reader.rs:944):
    (1) our footer parser on files written by Arrow C++ — schema, block count, and every Block must point at a
    framed message whose metadata length and body length agree with the message itself;
    (2) our footer builder under Arrow C++'s reader — a file assembled from pyarrow's own stream messages, our
    Block offsets and our trailer must read back batch for batch."""
    import ctypes as C
    import io
    import pyarrow as pa
    import arrow_rs_amd as A
    L = A._lib
    lib = L.load()
    sch = pa.schema([pa.field("a", pa.int64()), pa.field("b", pa.float64(), nullable=False), pa.field("s", pa.large_utf8()),
                     pa.field("t", pa.timestamp("us", "UTC")), pa.field("f", pa.bool_())])
    batches = [pa.record_batch([pa.array([1, None, 3], pa.int64()), pa.array([1.5, 2.5, 3.5]), pa.array(["x", None, "zz"], pa.large_utf8()),
                                pa.array([1, 2, None], pa.timestamp("us", "UTC")), pa.array([True, None, False])], schema=sch),
               pa.record_batch([pa.array([], pa.int64()), pa.array([], pa.float64()), pa.array([], pa.large_utf8()),
                                pa.array([], pa.timestamp("us", "UTC")), pa.array([], pa.bool_())], schema=sch),
               pa.record_batch([pa.array(list(range(100)), pa.int64()), pa.array([float(i) for i in range(100)]),
                                pa.array([str(i) for i in range(100)], pa.large_utf8()),
                                pa.array(list(range(100)), pa.timestamp("us", "UTC")), pa.array([i % 3 == 0 for i in range(100)])], schema=sch)]

    def parse(tail):
        flen = C.c_int64()
        st = lib.ah_ipc_decode_footer(None, tail, len(tail), C.byref(flen), None, None, None, None)
        if st != L.AH_OK:
            return st, None, None, None
        n, fields, nb, blocks = C.c_int32(), C.POINTER(L.IpcField)(), C.c_int32(), C.POINTER(L.IpcBlock)()
        st = lib.ah_ipc_decode_footer(None, tail, len(tail), C.byref(flen), C.byref(n), C.byref(fields), C.byref(nb), C.byref(blocks))
        if st != L.AH_OK:
            return st, None, None, flen.value
        f = [(fields[i].name.decode(), fields[i].format.decode(), bool(fields[i].nullable)) for i in range(n.value)]
        b = [(blocks[i].offset, blocks[i].meta_data_length, blocks[i].body_length) for i in range(nb.value)]
        lib.ah_host_free(fields)
        lib.ah_host_free(blocks)
        return st, f, b, flen.value

    # (1) Arrow C++ writes, we parse
    sink = io.BytesIO()
    with pa.ipc.new_file(sink, sch) as w:
        for b in batches:
            w.write_batch(b)
    data = sink.getvalue()
    st, fields, blocks, flen = parse(data[-10:])          # the last 10 bytes alone: footer length, then "need more"
    assert st == L.AH_PARSE_ERROR and flen == int.from_bytes(data[-10:-6], "little")
    st, fields, blocks, flen2 = parse(data[-(flen + 10):])
    assert st == L.AH_OK and flen2 == flen
    assert fields == [("a", "l", True), ("b", "g", False), ("s", "U", True), ("t", "tsu:UTC", True), ("f", "b", True)]
    assert len(blocks) == len(batches)
    for off, mlen, blen in blocks:
        assert data[off:off + 4] == b"\xff" * 4 and int.from_bytes(data[off + 4:off + 8], "little") == mlen - 8
        ht, bl = C.c_int32(), C.c_int64()
        msg = data[off:off + mlen]
        assert lib.ah_ipc_message_info(None, msg, len(msg), C.byref(ht), C.byref(bl)) == L.AH_OK
        assert (ht.value, bl.value) == (3, blen)
    # malformed trailers: the reference's texts (reader.rs:944-956)
    flen_c = C.c_int64()
    assert lib.ah_ipc_decode_footer(None, data[:-1], len(data) - 1, C.byref(flen_c), None, None, None, None) == L.AH_PARSE_ERROR
    assert lib.ah_last_error(None) in (b"no context", b"Arrow file does not contain correct footer")
    bad = data[:-10] + (-5).to_bytes(4, "little", signed=True) + b"ARROW1"
    assert lib.ah_ipc_decode_footer(None, bad, len(bad), C.byref(flen_c), None, None, None, None) == L.AH_PARSE_ERROR

    # (2) we build the trailer, Arrow C++ reads the file
    stream = io.BytesIO()
    with pa.ipc.new_stream(stream, sch) as w:
        for b in batches:
            w.write_batch(b)
    sdata = stream.getvalue()
    pos, msgs = 0, []
    while pos < len(sdata):  # walk [0xFFFFFFFF][len][metadata][body]
        assert sdata[pos:pos + 4] == b"\xff" * 4
        mlen = int.from_bytes(sdata[pos + 4:pos + 8], "little")
        if mlen == 0:
            break
        ht, bl = C.c_int32(), C.c_int64()
        m = sdata[pos:pos + 8 + mlen]
        assert lib.ah_ipc_message_info(None, m, len(m), C.byref(ht), C.byref(bl)) == L.AH_OK
        msgs.append((ht.value, pos, 8 + mlen, bl.value))
        pos += 8 + mlen + bl.value
    assert [m[0] for m in msgs] == [1, 3, 3, 3]
    header = b"ARROW1\x00\x00"
    bl = (L.IpcBlock * 3)()
    for i, (_, p, mlen, blen) in enumerate(msgs[1:]):
        bl[i].offset, bl[i].meta_data_length, bl[i].body_length = len(header) + p, mlen, blen
    arr = (L.IpcField * len(fields))()
    keep = [(nm.encode(), fm.encode()) for nm, fm, _ in fields]
    for i, (nm, fm) in enumerate(keep):
        arr[i].name, arr[i].format, arr[i].nullable = nm, fm, 1 if fields[i][2] else 0
    out, ln = C.c_void_p(), C.c_int64()
    assert lib.ah_ipc_file_footer(None, len(fields), arr, 3, bl, C.byref(out), C.byref(ln)) == L.AH_OK
    trailer = C.string_at(out, ln.value)
    lib.ah_host_free(out)
    assert trailer[-6:] == b"ARROW1" and int.from_bytes(trailer[-10:-6], "little") == ln.value - 10
    mine = header + sdata[:pos] + b"\xff\xff\xff\xff\x00\x00\x00\x00" + trailer
    rd = pa.ipc.open_file(pa.py_buffer(mine))
    assert rd.schema.equals(sch) and rd.num_record_batches == 3
    for i, b in enumerate(batches):
        assert rd.get_batch(i).equals(b)
    st, f2, b2, _ = parse(mine[-(ln.value):])            # and our parser on our own trailer
    assert st == L.AH_OK and f2 == fields and b2 == [(bl[i].offset, bl[i].meta_data_length, bl[i].body_length) for i in range(3)]


def test_integration_md_quotes_the_crate():
    """INTEGRATION.md section 2 shows the drop-in Rust layer by quoting bindings/rust/arrow-hip/src/kernels.rs VERBATIM
    (the blocks between its `// <<integration:NAME` ... `// integration>>` markers).  Neither compiles here, which is
    exactly why they must at least agree: this fails when one is edited without the other.  The
    quoted layer must carry the reference's own signatures."""
    src = open(os.path.join(ROOT, "bindings", "rust", "arrow-hip", "src", "kernels.rs")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = dict(re.findall(r"// <<integration:(\w+)\n(.*?)// integration>>", src, flags=re.S))
    assert set(blocks) >= {"owner", "view", "wrap", "filter", "take", "numeric", "cmp", "cast"}
    for name, text in blocks.items():
        assert text.strip() and text in md, f"INTEGRATION.md does not quote block `{name}` of kernels.rs verbatim"
    # the signatures north_star asks for (arrow-select/src/filter.rs:201, take.rs:89, arrow-arith/src/numeric.rs:41,
    # arrow-ord/src/cmp.rs:113, arrow-cast/src/cast/mod.rs:347)
    for sig in ("pub fn filter(values: &dyn Array, predicate: &BooleanArray) -> Result<ArrayRef, ArrowError>",
                "pub fn take(values: &dyn Array, indices: &dyn Array, options: Option<TakeOptions>) -> Result<ArrayRef, ArrowError>",
                "pub fn add_wrapping(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<ArrayRef, ArrowError>",
                "pub fn lt(lhs: &dyn Datum, rhs: &dyn Datum) -> Result<BooleanArray, ArrowError>",
                "pub fn cast(array: &dyn Array, to_type: &DataType) -> Result<ArrayRef, ArrowError>"):
        assert sig in src and sig in md, sig
    assert "Buffer::from_custom_allocation" in blocks["wrap"] and "NullBuffer::new_unchecked" in blocks["wrap"]
    # every sys:: name the drop-in layer uses exists in the generated declarations
    committed = open(os.path.join(ROOT, "bindings", "rust", "arrow-hip-sys", "src", "lib.rs")).read()
    known = set(re.findall(r"pub (?:fn|const|struct|type) (\w+)", committed))
    used = set(re.findall(r"sys::(\w+)", src))
    assert used <= known, sorted(used - known)
