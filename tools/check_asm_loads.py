#!/usr/bin/env python3
"""tools/check_asm_loads.py [file.hip ...] [-- hipcc flags] -- a static check of the kernels that issue their operand
loads by hand.

accum_mfma.hip and contract_mfma.hip fetch MFMA operands with inline-asm `global_load_dwordx2` and wait with exact
`s_waitcnt vmcnt(N)` (hipcc's own waitcnt placement exposes the full memory latency every trip).
Those loads are invisible to the compiler: between a load's issue and the wait that covers it, the destination register
still belongs to the load -- but the register allocator may think it free.  Round 4 lost a loop bound that way (the
prefetching wavefront's `sink` looked dead between two loads, the bound was put there and a late-returning fragment
overwrote it: tools/fuzz_parity.py case 40501).

This script compiles a source to gfx950 ISA (hipcc -S, no GPU needed), cuts every kernel into basic blocks and walks its
CONTROL FLOW with the hardware's rule -- vector memory operations complete in order, `s_waitcnt vmcnt(N)` returns when at
most N are outstanding -- reporting any instruction that READS or WRITES a vector register while a `global_load` into it
is still outstanding (a load into the register of an older outstanding load is fine: in-order return).

Round 5: the walk follows every edge, back edges included.  The state carried along an edge is the queue of outstanding
operations (destination registers, oldest first); a block is walked again for every queue it has not been entered with
before, so a loop body is walked with what its own bottom left outstanding -- a load issued at the end of one trip into a
register the top of the next trip touches (the shape of every operand ring in these kernels: fetch(next) ... arrive) is
seen -- and again until the queue at its head repeats.  Conditional branches go both ways; `s_endpgm` / `s_setpc` end a
path.  A block entered with more than MAX_STATES different queues stops being re-walked (reported as a note: not a
finding).  Exit 1 on any finding."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["accum_mfma.hip", "contract_mfma.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
MAX_STATES = 8192  # different queues a block is walked with (the shipped kernels need up to ~2000: partly drained rings)
MAX_QUEUE = 64    # the hardware counter holds 63


def vregs(tok):
    """vector registers named by one operand token: v12 -> {12}, v[4:7] -> {4,5,6,7}"""
    tok = tok.strip().split(" ")[0]
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return frozenset({int(m.group(1))})
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    return frozenset()


def union(sets):
    out = set()
    for s in sets:
        out |= s
    return out


LOAD = ("global_load", "buffer_load", "flat_load", "scratch_load")
STORE = ("global_store", "buffer_store", "flat_store", "scratch_store")


def compile_isa(src, extra=()):
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17", *extra, "-S",
                          "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True)
    if asm.returncode != 0:
        raise SystemExit("hipcc failed on %s:\n%s" % (src, asm.stderr[-2000:]))
    return asm.stdout


def kernels_of(isa):
    """-> [(name, [(line number, text, op, operands)], {label: index of the instruction it precedes})]"""
    out, cur = [], None
    for ln, raw in enumerate(isa.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if re.match(r"^_Z\w+:", line):
            cur = (line[:-1], [], {})
            out.append(cur)
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if line.endswith(":"):
            cur[2][line[:-1]] = len(cur[1])
            continue
        if line.startswith("."):
            continue
        op, _, rest = line.partition(" ")
        cur[1].append((ln, raw.strip(), op, [t.strip() for t in rest.split(",")] if rest.strip() else []))
    return out


def step(ins, fifo, findings, kernel):
    """one instruction against the queue of outstanding operations; returns the queue after it"""
    ln, raw, op, ops = ins
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", " ".join(ops))
        if m:
            keep = int(m.group(1))
            if len(fifo) > keep:
                fifo = fifo[len(fifo) - keep:] if keep else ()
        return fifo
    busy = union(d for _, d in fifo)
    if op.startswith(LOAD):
        srcs = union(vregs(t) for t in ops[1:])
        if srcs & busy:
            findings[(kernel, ln)] = (raw, "address register of a load is itself an outstanding load's destination")
        return (fifo + ((ln, vregs(ops[0]) if ops else frozenset()),))[-MAX_QUEUE:]
    if op.startswith(STORE) or "atomic" in op:
        used = union(vregs(t) for t in ops)
        if used & busy:
            findings[(kernel, ln)] = (raw, "store reads a register an outstanding load will write")
        return (fifo + ((ln, frozenset()),))[-MAX_QUEUE:]
    hit = union(vregs(t) for t in ops) & busy
    if hit:
        findings[(kernel, ln)] = (raw, "touches v%s while the load of line %d is outstanding"
                                  % (sorted(hit)[0], next(l for l, d in fifo if d & hit)))
    return fifo


def walk(name, code, labels, findings, notes):
    starts = set(labels.values()) | {0}
    seen = {}  # block start -> set of queue signatures it has been walked with
    work = [(0, ())]
    n_loads = sum(1 for c in code if c[2].startswith(LOAD))
    while work:
        i, fifo = work.pop()
        while i < len(code):
            if i in starts:
                sig = tuple(d for _, d in fifo)
                s = seen.setdefault(i, set())
                if sig in s:
                    break
                if len(s) >= MAX_STATES:
                    notes.add((name, code[i][0]))
                    break
                s.add(sig)
            ln, raw, op, ops = code[i]
            if op == "s_endpgm" or op.startswith("s_setpc"):
                break
            if op == "s_branch":
                t = labels.get(ops[0]) if ops else None
                if t is None:
                    break
                i = t
                continue
            if op.startswith("s_cbranch"):
                t = labels.get(ops[-1]) if ops else None
                if t is not None:
                    work.append((t, fifo))
                i += 1
                starts.add(i)  # (the fall-through is a block of its own)
                continue
            fifo = step(code[i], fifo, findings, name)
            i += 1
    return n_loads


def check(path, extra=()):
    src = path if os.path.isabs(path) else os.path.join(ROOT, "ngsdist_amd", "csrc", path)
    findings, notes, n_loads = {}, set(), 0
    for name, code, labels in kernels_of(compile_isa(src, extra)):
        n_loads += walk(name, code, labels, findings, notes)
    out = [(k, ln, raw, why) for (k, ln), (raw, why) in sorted(findings.items(), key=lambda kv: kv[0][1])]
    check.notes = sorted(notes)
    return out, n_loads


def main():
    args = sys.argv[1:]
    extra = ()
    if "--" in args:
        extra = tuple(args[args.index("--") + 1:])
        args = args[:args.index("--")]
    files = args or DEFAULT
    bad = 0
    for f in files:
        findings, n_loads = check(f, extra)
        print("%s%s: %d vector loads in the text, %d finding(s)%s"
              % (f, " " + " ".join(extra) if extra else "", n_loads, len(findings),
                 "; %d block(s) not walked with every queue (more than %d)" % (len(check.notes), MAX_STATES) if check.notes else ""))
        for k, ln, text, why in findings[:40]:
            print("  %s\n    line %d: %s\n    -> %s" % (k[:100], ln, text, why))
        bad += len(findings)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
