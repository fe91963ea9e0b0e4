#!/usr/bin/env python3
"""Build check: no kernel of libidsp_hip.so may use scratch memory unless it is on the allow-list.

Several kernels keep arrays in registers only because every loop over them is fully unrolled (`stage[NI]`, `ring[U]`,
`held[][]`, `p[LPT]`, `cur/nxt[NS]`); when an unroll fails those arrays silently move to scratch, and the Makefile
passes -Wno-pass-failed (the unroll warnings of the deliberately partially unrolled loops would drown everything else).
This script is the signal instead: it takes the gfx950 code objects out of the shared library, reads each kernel's
`.private_segment_fixed_size` / `.vgpr_spill_count` / `.sgpr_spill_count` from the code-object metadata and fails if a
kernel that is not listed in ALLOWED uses scratch or spills vector registers.

  python tools/check_scratch.py [--lib idsp_amd/lib/libidsp_hip.so] [--list]
"""
from __future__ import annotations

import argparse
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

# Kernels that are KNOWN to use scratch, with the reason (regular expressions on the demangled-ish symbol name).
ALLOWED = [
    # LaneMajor Cic decimator: a 4-tile register ring that spills 0.6-1.4 KB per thread; rings of 3 or 2 tiles without
    # spills measured slower at 16384 lanes (profiles/NOTES.md §7), left as it is
    (r"cic_dec_lm_kernel", "Cic LaneMajor decimator register ring (measured faster than the spill-free forms)"),
    # external-LO lock-in with [Lowpass<2>; 4] arms on the LaneMajor staged kernel, 64 lanes per wave (the form from 49152 lanes up, where the
    # launch is memory-bound): 46 VGPRs spill since the staged kernels' accesses name the global address space (round 6, tools/check_flat.py);
    # the 32- and 16-lane forms and every other lock-in processor stay in registers
    (r"stream_lane_major_staged.*LockinLoProcILi2ELi4EEELi64E", "68 B per thread since the staged kernels left flat accesses (round 6)"),
]


def kernels_of(lib: str):
    """[(name, scratch bytes, vgpr spills, sgpr spills, vgprs)] of every gfx950 kernel in `lib`."""
    tmp = tempfile.mkdtemp(prefix="idsp_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        # The library is linked with --offload-compress: .hip_fatbin holds one compressed bundle ("CCOB", version 3: magic,
        # u16 version, u16 method, u64 file size, u64 uncompressed size, u64 hash; bundles are padded to 4 KiB) per translation
        # unit.  Cut them apart and let clang-offload-bundler unpack the gfx950 code object of each.
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", local, fat], check=True)
        data = open(fat, "rb").read()
        pos, n = data.find(b"CCOB"), 0
        if pos < 0:  # not compressed: the plain route
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, cwd=tmp)
        while pos >= 0:
            size = struct.unpack("<Q", data[pos + 8:pos + 16])[0]
            piece = os.path.join(tmp, f"bundle{n}.bin")
            with open(piece, "wb") as fh:
                fh.write(data[pos:pos + size])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}",
                            f"--output={os.path.join(tmp, f'lib.so.{n}.gfx950')}"], check=True, capture_output=True)
            n += 1
            pos = data.find(b"CCOB", pos + size)
        out = []
        for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950"))):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                f = lambda key: re.search(r"\." + key + r":\s*(\S+)", blk)  # noqa: E731
                name = f("name")
                if not name:
                    continue
                out.append((name.group(1), int(f("private_segment_fixed_size").group(1)), int(f("vgpr_spill_count").group(1)),
                            int(f("sgpr_spill_count").group(1)), int(f("vgpr_count").group(1))))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "idsp_amd", "lib", "libidsp_hip.so"))
    ap.add_argument("--list", action="store_true", help="print every kernel with its scratch / spill figures")
    args = ap.parse_args(argv)
    ks = kernels_of(args.lib)
    if not ks:
        print("no gfx950 kernels found in", args.lib)
        return 2
    names = demangle([k[0] for k in ks])
    bad, allowed = [], []
    for (raw, scratch, vsp, ssp, vgprs), name in zip(ks, names):
        if args.list:
            print(f"{scratch:6d} B scratch  {vsp:4d} vgpr spills  {ssp:4d} sgpr spills  {vgprs:4d} vgprs  {name}")
        if scratch:  # (vgpr spills with 0 B of scratch went to AGPRs: no memory traffic)
            why = next((w for pat, w in ALLOWED if re.search(pat, name) or re.search(pat, raw)), None)
            (allowed if why else bad).append((name, scratch, vsp, why))
    print(f"{len(ks)} kernels, {len(allowed)} allowed scratch users, {len(bad)} unexpected")
    for name, scratch, vsp, why in allowed:
        print(f"  allowed: {scratch} B scratch, {vsp} vgpr spills: {name[:160]}  [{why}]")
    for name, scratch, vsp, _ in bad:
        print(f"  UNEXPECTED: {scratch} B scratch, {vsp} vgpr spills: {name[:300]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
