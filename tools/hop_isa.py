#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop, from the disassembly of the built library (no GPU needed).

usage: hop_isa.py [--lib PATH] [--dump] <substring of the demangled kernel name> ...

For every kernel whose demangled name contains the substring: the innermost-looking loops (backward branches), their instruction
counts by class (VALU split into float / integer / moves / selects / compares, SALU, LDS, VMEM, waits, branches) and the whole body.
The per-hop figures of profiles/r04_hop_isa.txt come from here: the hop loop of the narrow sweeps is the loop with the most LDS
instructions; its trip handles HOP_UNROLL hops of 64 * R slots."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CXXFILT = "c++filt"
HERE = os.path.dirname(os.path.abspath(__file__))

FLOAT = re.compile(r"^v_(add|sub|mul|min|max|fma|mac|fmac|cmp_class|cvt|rcp|div|ldexp|frexp|trunc|floor|ceil|rndne|pk_(add|mul|fma|min|max))_?.*(f16|f32|f64)")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_cndmask",)):
            return "valu_select"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
            return "valu_move"
        if op.startswith("v_cmp"):
            return "valu_cmp_f" if op.endswith(("f32", "f64", "f16")) or "class" in op else "valu_cmp_i"
        if FLOAT.match(op):
            return "valu_float"
        return "valu_int"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="hop_isa_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], check=True, capture_output=True, cwd=tmp)
        parts = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        out = ""
        for f in parts:
            out += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(text):
    """-> {mangled: [(addr, op, operands)]}"""
    funcs, cur = {}, None
    for l in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", l)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def demangle(names):
    out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def loops(ins):
    """backward branches -> [(start_index, end_index)] (end inclusive), innermost first by length"""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    res = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(-?\d+)\s*$", args.split()[-1]) if args else None
            # llvm-objdump prints the target as a label address in the comment-less form: "s_cbranch_scc1 65279" (simm16); compute the target
            if m:
                simm = int(m.group(1))
                if simm >= 0x8000:
                    simm -= 0x10000
                tgt = a + 4 + 4 * simm
                if tgt <= a and tgt in addr_to_idx:
                    res.append((addr_to_idx[tgt], i))
    return sorted(res, key=lambda r: r[1] - r[0])


def mix(ins):
    c = collections.Counter(classify(op) for _, op, _ in ins)
    c["total"] = len(ins)
    c["valu"] = sum(v for k, v in c.items() if k.startswith("valu_"))
    return c


def fmt(c):
    keys = ["total", "valu", "valu_float", "valu_int", "valu_select", "valu_move", "valu_cmp_i", "valu_cmp_f", "salu", "smem", "lds", "vmem", "wait", "branch", "other"]
    return "  ".join(f"{k}={c.get(k, 0)}" for k in keys)


def main():
    args = sys.argv[1:]
    lib = os.path.join(HERE, "..", "bdd_amd", "csrc", "libbdd_mma_hip.so")
    dump = False
    pats = []
    while args:
        a = args.pop(0)
        if a == "--lib":
            lib = args.pop(0)
        elif a == "--dump":
            dump = True
        else:
            pats.append(a)
    funcs = functions(disassemble(lib))
    names = demangle(list(funcs))
    for mangled, ins in funcs.items():
        dn = names[mangled]
        if not any(p in dn for p in pats):
            continue
        print(f"== {dn}")
        print(f"   whole kernel: {fmt(mix(ins))}")
        ls = loops(ins)
        # the hop loop: the loop with the most LDS instructions per instruction... report the three loops with the most LDS instructions
        # candidates: loops that hold LDS and VMEM instructions, shortest first (the hop loop is the shortest loop that stores);
        # backward s_cbranch_execz jumps into shared blocks show up as longer pseudo-loops behind it
        cand = [r for r in ls if any(op.startswith("ds_") for _, op, _ in ins[r[0]:r[1] + 1]) and any(op.startswith("buffer_store") for _, op, _ in ins[r[0]:r[1] + 1])]
        ranked = sorted(cand, key=lambda r: r[1] - r[0])[:3]
        for s, e in ranked:
            body = ins[s:e + 1]
            print(f"   loop @{ins[s][0]:#x}..{ins[e][0]:#x}: {fmt(mix(body))}")
            if dump:
                top = collections.Counter(op for _, op, _ in body).most_common(40)
                print("      " + ", ".join(f"{op} {n}" for op, n in top))
                for a, op, ar in body:
                    print(f"      {a:#x}  {op} {ar}")


if __name__ == "__main__":
    main()
