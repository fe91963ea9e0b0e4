"""The Rust host shim is source only (no cargo/rustc in the image), so what CAN be checked mechanically is checked:
  * rust/idsp-hip-sys/src/lib.rs is byte for byte what tools/gen_rust_sys.py generates from include/idsp_hip.h;
  * its `extern "C"` block declares exactly the header's symbols (symbol for symbol, same order) and every one of
    them is exported by the built libidsp_hip.so;
  * argument counts and pointer/scalar kinds of every Rust prototype agree with the ctypes prototypes the GPU tests
    call through (idsp_amd/_abi.py) — two independent transcriptions of the header;
  * every `sys::idsp_*` / `sys::Idsp*` / `sys::IDSP_*` the safe crate (rust/idsp-hip) uses exists in the sys crate;
  * `#[repr(C)]` struct sizes implied by the Rust field lists equal ctypes' sizes."""
import ctypes
import os
import re
import subprocess
import sys

from idsp_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "rust", "idsp-hip-sys", "src", "lib.rs")
SAFE = os.path.join(ROOT, "rust", "idsp-hip", "src", "lib.rs")


def header_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "idsp_hip.h")).read(), flags=re.S)
    return re.findall(r"\b(idsp_[a-z0-9_]+)\s*\(", text)


def rust_prototypes():
    src = open(SYS).read()
    block = src[src.index('unsafe extern "C" {'):]
    block = block[:block.index("\n}\n")]
    return re.findall(r"pub fn (idsp_\w+)\((.*?)\)(?: -> ([^;]+))?;", block)


def test_sys_crate_is_the_generated_file():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--stdout"], capture_output=True, text=True, check=True).stdout
    assert out == open(SYS).read(), "rust/idsp-hip-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py"


def test_extern_block_matches_header_symbol_for_symbol():
    protos = rust_prototypes()
    assert [p[0] for p in protos] == header_symbols()
    src = open(SYS).read()
    listed = re.findall(r'^    "(idsp_\w+)",$', src, flags=re.M)
    assert listed == header_symbols()
    from idsp_amd._lib import load

    _, lib = load()
    for name, _, _ in protos:
        assert hasattr(lib, name), name


def _kind(rust_ty):
    rust_ty = rust_ty.strip()
    if rust_ty.startswith("*"):
        return "ptr"
    if rust_ty == "idsp_shard_fn":
        return "ptr"
    return {"usize": "size", "c_int": "int", "f64": "f64", "u32": "u32", "i64": "i64", "i32": "int"}[rust_ty]


def _ckind(ct):
    if ct in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(ct, type) and issubclass(ct, ctypes._Pointer)):
        return "ptr"
    return {ctypes.c_size_t: "size", ctypes.c_int: "int", ctypes.c_double: "f64", ctypes.c_uint32: "u32", ctypes.c_int64: "i64"}[ct]


def test_rust_prototypes_agree_with_the_ctypes_prototypes():
    protos = {name: (args, ret) for name, args, ret in rust_prototypes()}
    table = {}
    for name, sig in list(_abi.PROCESSING.items()) + list(_abi.PITCHED.items()):
        table["idsp_" + name] = (list(sig), ctypes.c_int)
    for name, (res, args) in list(_abi.HELPERS.items()) + list(_abi.UTILS.items()) + list(_abi.FRONTEND.items()):
        table["idsp_" + name] = (list(args), res)
    assert sorted(table) == sorted(protos)
    for name, (cargs, cres) in table.items():
        rargs, rret = protos[name]
        rk = [_kind(a.split(":", 1)[1]) for a in rargs.split(", ")] if rargs else []
        assert rk == [_ckind(a) for a in cargs], name
        assert _kind(rret) == _ckind(cres), name


def test_safe_crate_only_uses_declared_items():
    sys_src = open(SYS).read()
    declared = set(re.findall(r"pub (?:fn|struct|const|type) (\w+)", sys_src))
    safe_src = open(SAFE).read()
    used = set(re.findall(r"\bsys::(\w+)", safe_src)) - {"idsp_"}  # "sys::idsp_*" appears in a doc comment
    # entry points handed to the kernel macros as bare identifiers (`sys::$entry`), and any other idsp_ symbol named
    used |= {t for t in re.findall(r"\b(idsp_[a-z0-9_]+)\b", safe_src) if not t.endswith("_")} - {"idsp_hip_sys", "idsp_hip", "idsp_status"}
    assert used and used <= declared, sorted(used - declared)
    for needed in ("idsp_biquad_i32_df1", "idsp_biquad_f32_df2t", "idsp_lockin_i32_process", "idsp_hbf_dec_f32", "idsp_device_alloc"):
        assert needed in used
    assert "impl<'a, 'b, C, S, L> SplitViewProcess<" in safe_src and "SplitViewInplace<" in safe_src


RUST_SIZES = {"i32": 4, "u32": 4, "f32": 4, "f64": 8, "i64": 8}


def _rust_sizeof(ty):
    ty = ty.strip()
    m = re.match(r"\[(.+); (\d+)\]$", ty)
    if m:
        return _rust_sizeof(m.group(1)) * int(m.group(2))
    return RUST_SIZES[ty] if ty in RUST_SIZES else STRUCTS[ty][0]


STRUCTS = {}


def test_repr_c_struct_sizes_match_ctypes():
    src = open(SYS).read()
    pairs = {"IdspBiquadI32": _abi.BiquadI32, "IdspBiquadClampI32": _abi.BiquadClampI32, "IdspBiquadF32": _abi.BiquadF32,
             "IdspBiquadClampF32": _abi.BiquadClampF32, "IdspBiquadF64": _abi.BiquadF64, "IdspBiquadClampF64": _abi.BiquadClampF64,
             "IdspHbfCascadeF32": _abi.HbfCascadeF32, "IdspFirSymF32": _abi.FirSymF32, "IdspHbfCascadeF64": _abi.HbfCascadeF64, "IdspFirSymF64": _abi.FirSymF64, "IdspLockinI32": _abi.LockinI32,
             "IdspWdf": _abi.Wdf, "IdspFmDisc": _abi.FmDisc, "IdspCic": _abi.Cic, "IdspFilter": _abi.Filter,
             "IdspPidBuilder": _abi.PidBuilder, "IdspUnits": _abi.Units, "IdspPid": _abi.Pid, "IdspBaConfig": _abi.BaConfig,
             "IdspFilterConfig": _abi.FilterConfig}
    found = [(n, b) for n, b in re.findall(r"pub struct (\w+) \{\n(.*?)\n\}", src, flags=re.S) if "_private" not in b]  # not the opaque handles
    assert sorted(n for n, _ in found) == sorted(pairs)
    for name, body in found:  # header order: nested structs are declared before their users
        size, align = 0, 1
        for ty in re.findall(r"pub \w+: (.+),", body):
            s = _rust_sizeof(ty)
            base = ty
            while base.startswith("["):  # element type of (nested) arrays
                base = base[1:base.rindex(";")]
            a = RUST_SIZES.get(base) or STRUCTS[base][1]
            size = (size + a - 1) // a * a + s
            align = max(align, a)
        size = (size + align - 1) // align * align
        STRUCTS[name] = (size, align)
        assert size == ctypes.sizeof(pairs[name]), name
