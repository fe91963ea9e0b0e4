#!/usr/bin/env python3
"""Generate rust/idsp-hip-sys/src/lib.rs — the raw `extern "C"` declarations of libidsp_hip.so — from
include/idsp_hip.h, so that the Rust binding can never drift from the C ABI: every `#define` constant, enum,
struct and function prototype of the header becomes one Rust item.  tests/test_rust_shim.py regenerates the file and
compares it with the committed one, and checks the symbol lists against each other.

  python tools/gen_rust_sys.py            # rewrite rust/idsp-hip-sys/src/lib.rs
  python tools/gen_rust_sys.py --stdout   # print instead

The header is plain C with a small vocabulary (stdint types, pointers, fixed arrays, no function pointers, no
bit-fields), which is all this parser understands; anything else raises."""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "idsp_hip.h")
OUT = os.path.join(ROOT, "rust", "idsp-hip-sys", "src", "lib.rs")

SCALARS = {
    "int": "c_int", "unsigned": "c_uint", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char",
    "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "int8_t": "i8", "uint8_t": "u8",
    "int16_t": "i16", "uint16_t": "u16", "void": "c_void",
}
KEYWORDS = {"type", "f32", "f64", "in", "ref", "fn", "mod", "use", "loop", "match", "move", "box", "priv", "self"}


def camel(name: str) -> str:
    """idsp_biquad_clamp_i32 -> IdspBiquadClampI32"""
    return "".join(p[:1].upper() + p[1:] for p in name.split("_"))


def ident(name: str) -> str:
    return name + "_" if name in KEYWORDS else name


def strip_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


FNPTR_TYPES: set[str] = set()


def rust_type(ctype: str, structs: set[str], enums: set[str]) -> str:
    """C declarator type (without the name / array suffix) -> Rust type."""
    t = ctype.strip()
    if t in FNPTR_TYPES:
        return t
    const = False
    if t.startswith("const "):
        const, t = True, t[6:].strip()
    parts = [q.strip() for q in t.split("*")]  # base, then the qualifier after each `*` ("" or "const")
    base, quals = parts[0], parts[1:]
    if base.endswith(" const"):
        const, base = True, base[:-6].strip()
    if base in SCALARS:
        r = SCALARS[base]
    elif base in structs:
        r = camel(base)
    elif base in enums:
        r = "c_int"
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    for i in range(len(quals)):
        # pointee constness: the declared base constness for the innermost level, the qualifier written after the
        # previous `*` for the outer ones (`void *const *p` = pointer to const pointer to void)
        pointee_const = const if i == 0 else quals[i - 1] == "const"
        r = ("*const " if pointee_const else "*mut ") + r
    return r


def parse(text: str):
    text = strip_comments(text)
    defines = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(IDSP_[A-Z0-9_]+)\s+(-?\d+)\s*$", text, flags=re.M)]
    enums = []
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        items, nxt = [], 0
        for it in m.group(2).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                k, v = [s.strip() for s in it.split("=")]
                nxt = int(v)
            else:
                k = it
            items.append((k, nxt))
            nxt += 1
        enums.append((m.group(3), items))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            # `double t, x, y` style lists
            head = decl.split(",")
            first = head[0].rsplit(" ", 1)
            ctype = first[0]
            for nm in [first[1]] + [h.strip() for h in head[1:]]:
                dims = [int(d) if d.isdigit() else d for d in re.findall(r"\[(\w+)\]", nm)]
                fields.append((ctype, re.sub(r"\[.*", "", nm), dims))
        structs.append((m.group(3), fields))
    opaque = [m.group(2) for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text)]
    fnptrs = []
    for m in re.finditer(r"typedef\s+(\w[\w\s\*]*?)\(\s*\*\s*(\w+)\s*\)\s*\(([^()]*)\)\s*;", text):
        args = []
        for a in " ".join(m.group(3).split()).split(","):
            mm = re.match(r"(.*?)(\w+)$", a.strip())
            args.append((mm.group(1).strip().replace(" *", "*"), mm.group(2)))
        fnptrs.append((m.group(2), " ".join(m.group(1).split()), args))
    body = re.sub(r"typedef\s+(enum|struct)\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", " ", body)
    body = re.sub(r"typedef\s+\w[\w\s\*]*?\(\s*\*\s*\w+\s*\)\s*\([^()]*\)\s*;", " ", body)
    body = re.sub(r"^\s*#.*$", " ", body, flags=re.M)          # preprocessor lines
    body = body.replace('extern "C" {', " ")
    funcs = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(idsp_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[(\d*)\]$", a)
                if arr:  # `const double sos[6]` decays to a pointer
                    a = a[: arr.start()].strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                if arr:
                    ctype += " *"
                params.append((ctype.replace(" *", "*").replace("* ", "*"), pname))
        funcs.append((ret.replace(" *", "*"), name, params))
    return defines, enums, structs, funcs, opaque, fnptrs


def generate() -> str:
    text = open(HEADER).read()
    defines, enums, structs, funcs, opaque, fnptrs = parse(text)
    dvals = dict(defines)
    snames = {s for s, _ in structs} | set(opaque)
    enames = {e for e, _ in enums}
    FNPTR_TYPES.update(f for f, _, _ in fnptrs)
    out = []
    w = out.append
    w("//! Raw FFI declarations of `libidsp_hip.so` (include/idsp_hip.h) — the MI355X bulk engine for the per-sample")
    w("//! filter hot path of `idsp`.  GENERATED by tools/gen_rust_sys.py from the C header; do not edit.")
    w("//!")
    w("//! The reference workspace forbids `unsafe` (Cargo.toml:19 of quartiq/idsp), so this crate lives outside it.")
    w("#![no_std]")
    w("#![allow(non_camel_case_types, non_upper_case_globals, unused_imports, clippy::too_many_arguments)]")
    w("")
    w("use core::ffi::{c_char, c_int, c_uint, c_void};")
    w("")
    for k, v in defines:
        w(f"pub const {k}: usize = {v};" if not v.startswith("-") else f"pub const {k}: c_int = {v};")
    w("")
    for ename, items in enums:
        w(f"/// `{ename}` (C enum, passed as `int`)")
        for k, v in items:
            w(f"pub const {k}: c_int = {v};")
        w("")
    for sname, fields in structs:
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug, PartialEq)]")
        w(f"pub struct {camel(sname)} {{")
        for ctype, fname, dims in fields:
            t = rust_type(ctype, snames, enames)
            for d in reversed(dims):
                n = d if isinstance(d, int) else int(dvals[d])
                t = f"[{t}; {n}]"
            w(f"    pub {ident(fname)}: {t},")
        w("}")
        w("")
    for oname in opaque:
        w(f"/// Opaque `{oname}` handle (only ever used behind a pointer).")
        w("#[repr(C)]")
        w(f"pub struct {camel(oname)} {{")
        w("    _private: [u8; 0],")
        w("}")
        w("")
    for fname, ret, args in fnptrs:
        ps = ", ".join(f"{ident(p)}: {rust_type(t, snames, enames)}" for t, p in args)
        w(f"/// C callback type `{fname}`")
        w(f"pub type {fname} = Option<unsafe extern \"C\" fn({ps}) -> {rust_type(ret, snames, enames)}>;")
        w("")
    w('#[link(name = "idsp_hip")]')
    w('unsafe extern "C" {')
    for ret, name, params in funcs:
        ps = ", ".join(f"{ident(p)}: {rust_type(t, snames, enames)}" for t, p in params)
        r = "" if ret == "void" else f" -> {rust_type(ret, snames, enames)}"
        w(f"    pub fn {name}({ps}){r};")
    w("}")
    w("")
    w("/// Every symbol declared above, in header order (tests/test_rust_shim.py checks it against the C header and")
    w("/// against the exports of the built library).")
    w(f"pub const SYMBOLS: [&str; {len(funcs)}] = [")
    for _, name, _ in funcs:
        w(f'    "{name}",')
    w("];")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    src = generate()
    if "--stdout" in sys.argv:
        sys.stdout.write(src)
    else:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        with open(OUT, "w") as f:
            f.write(src)
        print(f"wrote {OUT} ({src.count(chr(10))} lines)")
