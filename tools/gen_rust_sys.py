"""Generate bindings/rust/arrow-hip-sys/src/lib.rs — the 1:1 Rust declarations of include/arrow_hip.h.

No Rust toolchain exists in this image, so the output has never been compiled here; generating it mechanically
from the header (and checking in tests/test_abi_symbols.py that it is up to date and covers every exported
symbol) is what keeps it honest.  The safe wrapper crate next to it (bindings/rust/arrow-hip) is hand-written.

    python tools/gen_rust_sys.py            # rewrite the file
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "arrow_hip.h")
OUT = os.path.join(ROOT, "bindings", "rust", "arrow-hip-sys", "src", "lib.rs")

BASE = {
    "void": "c_void", "char": "c_char", "int": "c_int", "double": "f64", "float": "f32", "size_t": "usize",
    "int8_t": "i8", "int16_t": "i16", "int32_t": "i32", "int64_t": "i64",
    "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64",
    # i32 aliases of the header
    "ah_status": "ah_status", "ah_type": "ah_type", "ah_arith_op": "ah_arith_op", "ah_cmp_op": "ah_cmp_op",
    "ah_boolean_op": "ah_boolean_op", "ah_agg_op": "ah_agg_op", "ah_like_op": "ah_like_op", "ah_time_unit": "ah_time_unit", "ArrowDeviceType": "ArrowDeviceType",
    # opaque / struct types keep their names
    "ah_context": "ah_context", "ah_filter_predicate": "ah_filter_predicate", "ah_array_view": "ah_array_view",
    "ah_array_out": "ah_array_out", "ah_scalar": "ah_scalar", "ah_data_type": "ah_data_type", "ah_ipc_field": "ah_ipc_field", "ah_ipc_block": "ah_ipc_block",
    "ArrowArray": "ArrowArray", "ArrowSchema": "ArrowSchema", "ArrowDeviceArray": "ArrowDeviceArray",
    "ah_alloc_fn": "ah_alloc_fn", "ah_free_fn": "ah_free_fn",
    "ah_comm": "ah_comm", "ah_exchange_stats": "ah_exchange_stats", "ah_context_stats_t": "ah_context_stats_t", "ah_coalescer": "ah_coalescer", "ah_coalescer_push": "ah_coalescer_push", "ah_graph": "ah_graph", "ah_exchange": "ah_exchange", "ah_filter_term": "ah_filter_term",
}
RUST_KEYWORDS = {"type", "ref", "in", "fn", "move", "match", "loop", "box", "use", "mod", "impl", "self", "where"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(ctype):
    """'const ah_array_view*' -> '*const ah_array_view'; 'uint8_t**' -> '*mut *mut u8'; 'int32_t' -> 'i32'."""
    t = ctype.replace("struct ", "").strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    names = [x for x in t if x != "const"]
    assert len(names) == 1, ctype
    base = BASE[names[0]]
    if stars == 0:
        return base
    # only the innermost pointer carries the C const
    out = ("*const " if const else "*mut ") + base
    for _ in range(stars - 1):
        out = "*mut " + out
    return out


def split_param(p):
    """'const ah_array_view* values' -> (ctype, name); arrays 'uint8_t bytes[32]' are handled by the struct code."""
    p = p.strip()
    if p.endswith("*"):  # unnamed pointer parameter: 'struct ArrowArray*'
        return p, None
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
    ctype, name = m.group(1).strip(), m.group(2)
    if not ctype:  # unnamed parameter such as 'void'
        return name, None
    return ctype, name


def ident(name):
    return name + "_" if name in RUST_KEYWORDS else name


def parse(text):
    text = strip_comments(text)
    defines = re.findall(r"#define\s+(ARROW_DEVICE_[A-Z_]+)\s+(\d+)", text)
    text = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("#"))  # drop the preprocessor lines
    consts = []
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for item in body.split(","):
            item = item.strip()
            if item:
                k, v = [x.strip() for x in item.split("=")]
                consts.append((k, v))
    consts.extend(defines)
    aliases = re.findall(r"typedef\s+int32_t\s+([A-Za-z_]+)\s*;", text)
    structs = []
    for m in re.finditer(r"(?:typedef\s+)?struct\s+([A-Za-z_]+)\s*\{(.*?)\}\s*([A-Za-z_]*)\s*;", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for line in body.split(";"):
            line = " ".join(line.split())
            if not line:
                continue
            fp = re.match(r"^(.*?)\(\*([a-z_]+)\)\((.*)\)$", line)  # function-pointer member (ArrowArray.release ...)
            if fp:
                ret, fname, params = fp.group(1).strip(), fp.group(2), fp.group(3)
                ptypes = ", ".join(rust_type(split_param(x)[0]) for x in params.split(","))
                r = "" if ret == "void" else " -> " + rust_type(ret)
                fields.append((fname, f"Option<unsafe extern \"C\" fn({ptypes}){r}>"))
                continue
            arr = re.match(r"^(.*?)([a-z_]+)\[(\d+)\]$", line)
            if arr:
                fields.append((arr.group(2), f"[{rust_type(arr.group(1))}; {arr.group(3)}]"))
                continue
            ctype, fname = split_param(line)
            fields.append((fname, rust_type(ctype)))
        structs.append((name, fields))
    funcs = []
    for m in re.finditer(r"AH_API\s+(.*?)\s*\b(ah_[a-z0-9_]+)\s*\((.*?)\)\s*;", text, flags=re.S):
        ret, name, params = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        args = []
        if params != "void":
            for p in params.split(","):
                ctype, pname = split_param(p)
                args.append((ident(pname), rust_type(ctype)))
        funcs.append((name, args, None if ret == "void" else rust_type(ret)))
    return consts, aliases, structs, funcs


def render(consts, aliases, structs, funcs):
    o = []
    o.append("//! arrow-hip-sys — raw declarations of `include/arrow_hip.h` (libarrow_hip.so, gfx950).")
    o.append("//!")
    o.append("//! GENERATED by tools/gen_rust_sys.py from the C header; do not edit.  Never compiled in the image this")
    o.append("//! repository is developed in (it has no Rust toolchain): treat as the binding a maintainer would start from.")
    o.append("#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case, clippy::missing_safety_doc)]")
    o.append("use core::ffi::{c_char, c_int, c_void};")
    o.append("")
    for a in aliases:
        o.append(f"pub type {a} = i32;")
    o.append("")
    for k, v in consts:
        o.append(f"pub const {k}: i32 = {v};")
    o.append("")
    o.append("/// Opaque: one HIP stream + pooled HBM allocator.")
    o.append("#[repr(C)] pub struct ah_context { _private: [u8; 0] }")
    o.append("/// Opaque: FilterPredicate (mask + device prefix tables).")
    o.append("#[repr(C)] pub struct ah_filter_predicate { _private: [u8; 0] }")
    o.append("/// Opaque: one rank of the multi-GPU exchange (an RCCL communicator).")
    o.append("#[repr(C)] pub struct ah_comm { _private: [u8; 0] }")
    o.append("pub const AH_COMM_ID_BYTES: usize = 128;")
    o.append("/// Opaque: BatchCoalescer state (in-progress buffers + completed queue).")
    o.append("#[repr(C)] pub struct ah_coalescer { _private: [u8; 0] }")
    o.append("#[repr(C)] pub struct ah_coalescer_push { _private: [u8; 0] }")
    o.append("/// Opaque: a recorded sequence of deferred calls (hipGraph + its executable).")
    o.append("#[repr(C)] pub struct ah_graph { _private: [u8; 0] }")
    o.append("#[repr(C)] pub struct ah_exchange { _private: [u8; 0] }")
    o.append("pub type ah_alloc_fn = Option<unsafe extern \"C\" fn(user: *mut c_void, bytes: usize) -> *mut c_void>;")
    o.append("pub type ah_free_fn = Option<unsafe extern \"C\" fn(user: *mut c_void, ptr: *mut c_void, bytes: usize)>;")
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Debug, Clone, Copy)]")
        o.append(f"pub struct {name} {{")
        for fname, ftype in fields:
            o.append(f"    pub {ident(fname)}: {ftype},")
        o.append("}")
        o.append("")
    o.append("#[link(name = \"arrow_hip\")]")
    o.append("extern \"C\" {")
    for name, args, ret in funcs:
        a = ", ".join(f"{n}: {t}" for n, t in args)
        r = f" -> {ret}" if ret else ""
        o.append(f"    pub fn {name}({a}){r};")
    o.append("}")
    return "\n".join(o) + "\n"


def generate():
    with open(HEADER) as f:
        return render(*parse(f.read()))


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        with open(OUT) as f:
            sys.exit(0 if f.read() == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
