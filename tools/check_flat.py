#!/usr/bin/env python3
"""Build check: no kernel of libidsp_hip.so may access memory through FLAT instructions.

A pointer that went through an integer (a wave-uniform base pinned to SGPRs, `uniform_ptr`) is a generic one, and hipcc then
emits `flat_load_* / flat_store_*`.  On gfx9 those count on lgkmcnt as well as on vmcnt: a 4-bit counter, so a wave has at most
15 of them in flight, and every `s_waitcnt lgkmcnt` of the wave's LDS traffic waits for its global traffic too.  Round 6 found
both staged stream kernels (`stream_lane_major_staged`, the C2 LaneMajor kernel, and `stream_frame_major_staged`) on flat
accesses; they now name the address space (`global_ld / global_st`, common.h).  This script disassembles every gfx950 code
object of the library and fails if a kernel contains a flat access.

  python tools/check_flat.py [--lib idsp_amd/lib/libidsp_hip.so] [--list]
"""
from __future__ import annotations

import argparse
import collections
import glob
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


# Kernels that may keep flat accesses, with the reason (regular expressions on the demangled name).
ALLOWED = [
    # generic-taps half-band interpolator (custom taps only; the built-in cascades run on hbf_wave.h): its stage body stores through a
    # pointer that is y for the last stage and the next stage's LDS stream otherwise
    (r"hbf_int_kernel", "stage output pointer is LDS or global by stage"),
]


def code_objects(lib: str, tmp: str):
    local = os.path.join(tmp, "lib.so")
    shutil.copy(lib, local)
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", local, fat], check=True)
    data = open(fat, "rb").read()
    pos, n = data.find(b"CCOB"), 0  # compressed bundles, one per translation unit (tools/check_scratch.py)
    while pos >= 0:
        size = struct.unpack("<Q", data[pos + 8:pos + 16])[0]
        piece = os.path.join(tmp, f"bundle{n}.bin")
        with open(piece, "wb") as fh:
            fh.write(data[pos:pos + size])
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={piece}", f"--output={os.path.join(tmp, f'co{n}.gfx950')}"], check=True, capture_output=True)
        n += 1
        pos = data.find(b"CCOB", pos + size)
    return sorted(glob.glob(os.path.join(tmp, "co*.gfx950")))


def flat_users(lib: str):
    """{kernel symbol: Counter of flat mnemonics}"""
    tmp = tempfile.mkdtemp(prefix="idsp_flat_")
    out = {}
    try:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1)
                    continue
                m = re.match(r"^\s+(flat_(?:load|store|atomic)\w*)", line)
                if m and cur:
                    out.setdefault(cur, collections.Counter())[m.group(1)] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "idsp_amd", "lib", "libidsp_hip.so"))
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    users = flat_users(a.lib)
    names = sorted(users)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    bad = 0
    for n, d in zip(names, dem):
        why = next((w for rx, w in ALLOWED if re.search(rx, d)), None)
        if why is None:
            bad += 1
        if why is None or a.list:
            print(f"{'FLAT ' if why is None else 'allow'}  {dict(users[n])}  {d[:200]}" + (f"   [{why}]" if why else ""))
    print(f"{len(names)} kernels with flat accesses, {bad} not on the allow-list")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
