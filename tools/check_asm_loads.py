#!/usr/bin/env python3
"""tools/check_asm_loads.py [file.hip ...] -- a static check of the kernels that issue their operand loads by hand.

accum_mfma.hip and contract_mfma.hip fetch MFMA operands with inline-asm `global_load_dwordx2` and wait with exact
`s_waitcnt vmcnt(N)` (hipcc's own waitcnt placement exposes the full memory latency every trip).
Those loads are invisible to the compiler: between a load's issue and the wait that covers it, the destination register
still belongs to the load -- but the register allocator may think it free.  Round 4 lost a loop bound that way (the
prefetching wavefront's `sink` looked dead between two loads, the bound was put there and a late-returning fragment
overwrote it: tools/fuzz_parity.py case 40501).

This script compiles a source to gfx950 ISA (hipcc -S, no GPU needed) and walks every kernel in program order with the
hardware's rule -- vector memory operations complete in order, `s_waitcnt vmcnt(N)` returns when at most N are outstanding --
and reports any instruction that READS or WRITES a vector register while a `global_load` into it is still outstanding
(a load into the register of an older outstanding load is fine: in-order return).  Program order is not control flow: a
loop's back edge is not followed, a conditional branch is taken to fall through, and the text behind an unconditional
branch starts from "nothing outstanding"; for these kernels (straight-line trips, waits at the head of every trip) that is
the useful approximation -- it finds the case above (git show 8c2bcc2^:ngsdist_amd/csrc/accum_mfma.hip) and passes the
fixed source.  Exit 1 on any finding."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["accum_mfma.hip", "contract_mfma.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def vregs(tok):
    """vector registers named by one operand token: v12 -> {12}, v[4:7] -> {4,5,6,7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(path, extra=()):
    src = path if os.path.isabs(path) else os.path.join(ROOT, "ngsdist_amd", "csrc", path)
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17", *extra, "-S",
                          "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True)
    if asm.returncode != 0:
        raise SystemExit("hipcc failed on %s:\n%s" % (src, asm.stderr[-2000:]))
    findings = []
    kernel, fifo, n_loads = None, [], 0  # fifo: outstanding vector memory operations, oldest first: (line, dest registers)
    for ln, raw in enumerate(asm.stdout.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if re.match(r"^_Z\w+:", line):
            kernel, fifo = line[:-1], []
            continue
        if kernel is None or line.startswith(".") or line.endswith(":"):
            continue
        if line.startswith("s_endpgm") or line.startswith("s_branch") or line.startswith("s_setpc"):
            fifo = []  # what follows in the text is reached by jumps only: not this path's state
            continue
        op, _, rest = line.partition(" ")
        ops = [t.strip() for t in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                fifo = fifo[len(fifo) - keep:] if keep and len(fifo) > keep else ([] if not keep else fifo)
            continue
        busy = set().union(*[d for _, d in fifo]) if fifo else set()
        if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
            n_loads += 1
            srcs = set().union(*[vregs(t) for t in ops[1:]]) if len(ops) > 1 else set()
            if srcs & busy:
                findings.append((kernel, ln, raw.strip(), "address register of a load is itself an outstanding load's destination"))
            fifo.append((ln, vregs(ops[0]) if ops else set()))
            continue
        if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or "atomic" in op:
            used = set().union(*[vregs(t) for t in ops]) if ops else set()
            if used & busy:
                findings.append((kernel, ln, raw.strip(), "store reads a register an outstanding load will write"))
            fifo.append((ln, set()))
            continue
        touched = set().union(*[vregs(t.split(" ")[0]) for t in ops]) if ops else set()
        hit = touched & busy
        if hit:
            findings.append((kernel, ln, raw.strip(), "touches v%s while the load of line %d is outstanding"
                             % (sorted(hit)[0], next(l for l, d in fifo if d & hit))))
    return findings, n_loads


def main():
    files = sys.argv[1:] or DEFAULT
    bad = 0
    for f in files:
        findings, n_loads = check(f)
        print("%s: %d vector loads walked, %d finding(s)" % (f, n_loads, len(findings)))
        for k, ln, text, why in findings[:40]:
            print("  %s\n    line %d: %s\n    -> %s" % (k[:100], ln, text, why))
        bad += len(findings)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
