#!/usr/bin/env python3
"""Instruction histogram of a kernel's steady-state loop in hipcc's -save-temps assembly.

usage: isa_hist.py FILE.s KERNEL_SUBSTRING [--all]
Finds the kernel whose mangled name contains the substring, takes the span from the first to the last line the compiler
marked `in Loop: Header=...` (plus the header block), and counts mnemonics.  --all: the whole kernel instead."""
import collections
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    whole = "--all" in sys.argv
    s = open(path).read()
    names = [m.group(1) for m in re.finditer(r"^(\S+):\s*; @\S+", s, re.M)]
    name = next(n for n in names if sub in n)
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    if not whole:
        hdrs = collections.Counter(re.findall(r"in Loop: Header=(\S+)", "\n".join(body)))
        hdr = hdrs.most_common(1)[0][0]
        lines = [k for k, l in enumerate(body) if f"Header={hdr} " in l or l.startswith(f".L{hdr}:")]
        body = body[min(lines):max(lines) + 40]
        # extend to the back edge
        for k, l in enumerate(body[::-1]):
            if f".L{hdr}" in l and ("s_branch" in l or "s_cbranch" in l):
                body = body[:len(body) - k]
                break
    ops = collections.Counter()
    for l in body:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m:
            ops[m.group(1)] += 1
    tot = sum(ops.values())
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    lds = sum(v for k, v in ops.items() if k.startswith("ds_"))
    print(f"{name[:80]}: {tot} instructions, {valu} VALU, {lds} LDS")
    for k, v in ops.most_common():
        print(f"  {k:28s} {v}")
    meta = s[s.index(".amdhsa_kernel " + name):]
    for key in ("next_free_vgpr", "next_free_sgpr"):
        m = re.search(rf"\.amdhsa_{key} (\d+)", meta)
        print(key, m.group(1) if m else "?")


if __name__ == "__main__":
    main()
