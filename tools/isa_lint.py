#!/usr/bin/env python3
"""ISA lint of a built library: no unpadded VMEM-store write-data hazard (run by the Makefile after linking and by tests/test_isa_lint.py).

On gfx950 a VMEM store of more than 64 bits (buffer_store_dwordx3 / x4) needs one wait state before a VALU instruction overwrites its
data registers.  hipcc pads the pair when the store's soffset is an immediate, but NOT when it is an SGPR — the ISA manuals exempt that
form and LLVM's hazard recogniser follows them — and the hardware then stores the overwritten dword in ~1 % of the cases
(tools/store_hazard.hip; profiles/r03_exchange_variant_rootcause.txt).  The kernels guard their 16-byte scalar-offset stores themselves
(kernels.hpp: hop_store(double2)); this disassembles the device code and reports every wide buffer store with an SGPR soffset that is
followed directly by a VALU write of one of its data registers.

The same check covers global_store_dwordx3 / x4 with an SGPR base (`saddr`), the form the 64-bit `big` paths can emit.

usage: isa_lint.py <library.so>     exit code 1 on a hit; 0 (with a note) when llvm-objdump is not installed; 3 when the
disassembly itself failed (the Makefile removes the library on ANY non-zero exit; `make LINT_OPTIONAL=1` keeps it on exit 3 only)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*\w+,\s*s\[\d+:\d+\],\s*(s\d+|m0|\d+|0x[0-9a-f]+)\b")
GSTORE = re.compile(r"^\s*global_store_dwordx[34]\s+v(?:\d+|\[\d+:\d+\]),\s*v\[(\d+):(\d+)\],\s*(s\[\d+:\d+\]|off)\b")
VALU_DST = re.compile(r"^\s*(v_[a-z0-9_]+)\s+(?:v(\d+)|v\[(\d+):(\d+)\])\b")


def disassemble(lib_path, tmp):
    lib = os.path.join(tmp, "lib.so")
    shutil.copy(lib_path, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=tmp)
    parts = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
    assert parts, "no device code object in the library"
    text = []
    for f in parts:
        text += subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout.splitlines()
    return text


def lint(lib_path):
    """-> (number of wide stores with an SGPR soffset, [(function, store, following VALU write)], number of buffer stores seen)"""
    with tempfile.TemporaryDirectory() as tmp:
        lines = disassemble(lib_path, tmp)
    func, wide_sgpr, hits = "?", 0, []
    pending = None  # (lo, hi, line) of a wide store with an SGPR soffset whose next instruction has not been seen yet
    for l in lines:
        if l.endswith(">:"):
            func, pending = l.split("<")[-1][:-2], None
            continue
        ins = l.split("//")[0]
        if not ins.strip() or not ins.startswith("\t"):
            continue
        if pending is not None:
            m = VALU_DST.match(ins)
            if m and not m.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                lo = int(m.group(2)) if m.group(2) is not None else int(m.group(3))
                hi = int(m.group(2)) if m.group(2) is not None else int(m.group(4))
                if not (hi < pending[0] or lo > pending[1]):
                    hits.append((func, pending[2].strip(), ins.strip()))
            pending = None
        m = STORE.match(ins) or GSTORE.match(ins)
        if m and m.group(3).startswith(("s", "m")):
            wide_sgpr += 1
            pending = (int(m.group(1)), int(m.group(2)), ins)
    return wide_sgpr, hits, sum(1 for l in lines if "buffer_store_dword" in l or "global_store_dword" in l)


if __name__ == "__main__":
    if not os.path.exists(OBJDUMP):
        print("[isa_lint] llvm-objdump not found: skipped")
        sys.exit(0)
    try:
        n, hits, stores = lint(sys.argv[1])
    except (subprocess.CalledProcessError, OSError, AssertionError) as e:
        print(f"[isa_lint] WARNING: could not disassemble {sys.argv[1]} ({e}): NOT linted")
        sys.exit(3)
    if hits:
        print(f"[isa_lint] {len(hits)} unpadded wide store / VALU-write pairs, e.g. {hits[:3]}")
        sys.exit(1)
    print(f"[isa_lint] {stores} buffer / global stores, {n} wide ones with an SGPR offset or base, all padded")
