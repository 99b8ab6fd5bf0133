#!/usr/bin/env python3
"""Static check of the kernels that issue their text loads from inline asm and wait for them with hand-counted s_waitcnt.

Between such a load and the wait that covers it the destination registers belong to the memory system, and the compiler
does not know: if register allocation moves a tile there (a live-range split, a spill) the move reads what has not
arrived -- silently, and only when the data is late.  (Found in round 4 in the first form of CountingRowKernel.)

The check walks each kernel's instructions in layout order: a global_load_dwordx4 puts its destination registers on the
in-flight list, s_waitcnt vmcnt(N) retires all but the N youngest loads (loads return in order), and any other instruction
that names an in-flight register is reported.  Layout order is not execution order: loops are handled by walking the
body twice (the state at the back edge is carried round once), forward branches are ignored -- so a report is a place to
look at, and no report is what the kernels rely on.

usage: inflight_registers.py <unit.hip> <kernel-name-substring> [...]"""
import re
import subprocess
import sys

HIPCC = "/opt/rocm/bin/hipcc"


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name, body = m.group(1), []
        elif line.startswith(".Lfunc_end") and name:
            yield name, body
            name = None
        elif name is not None:
            body.append(line)


def check(body):
    """[(line number, text)] of instructions touching a register a load still owes data to."""
    reports = []
    loops = []          # (label line index) of loop headers seen, for the second pass
    inflight = []       # list of register sets, oldest first
    labels = {}
    code = []
    from_asm = []       # code[i] stands inside an inline-asm statement (between ;;#ASMSTART and ;;#ASMEND)
    in_asm = False
    has_markers = any("#ASMSTART" in line for line in body)
    for line in body:
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        text = line.split(";")[0].strip()
        code.append(text)
        from_asm.append(in_asm or not has_markers)   # (hand-made listings of the tests carry no markers: every load counts)
        m = re.match(r"^(\.LBB\S+):", text)
        if m:
            labels[m.group(1)] = len(code) - 1

    def step(i, text):
        nonlocal inflight
        if not text or text.endswith(":") or text.startswith("."):
            return
        op = text.split()[0]
        if op.startswith("global_load") or op.startswith("scratch_load") or op.startswith("buffer_load"):
            dst = text.split()[1].rstrip(",")
            used = regs(" ".join(text.split()[2:]))
            for r in inflight:
                if r & (used | regs(dst)):
                    reports.append((i, text))
            # only the loads issued from INLINE ASM are the text tiles' (waits counted by hand: the compiler does not know they
            # are outstanding); every other load is the compiler's own -- offsets, table entries, end-of-string records --,
            # which it waits for itself before it names their registers again: they take a place in the queue and owe
            # nothing here (layout order is not execution order: the two arms of a branch follow each other in the
            # listing, and a register loaded in one and cleared in the other looked like a touched tile -- round 5)
            inflight.append(regs(dst) if op.startswith("global_load_dwordx4") and from_asm[i] else set())
            return
        if op.startswith("global_store") or op.startswith("scratch_store") or op.startswith("global_atomic"):
            inflight.append(set())      # counts in vmcnt, owes nothing
            for r in inflight:
                if r & regs(text):
                    reports.append((i, text))
            return
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n else []
            return
        touched = regs(text)
        for r in inflight:
            if r & touched:
                reports.append((i, text))
                break

    # the window loop: the innermost back edge whose body holds at least two groups of eight line loads (one per role of
    # the two tiles) -- or one group, for a kernel with one tile; everything else of the kernel (staging, flush) uses
    # compiler-generated loads, which the compiler waits for itself
    best = None
    for i, text in enumerate(code):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", text)
        if not m:
            continue
        target = labels.get(m.group(1) or m.group(2))
        if target is None or target > i:
            continue
        n = sum(1 for t in code[target:i] if t.startswith("global_load_dwordx4"))
        if n >= 8 and (best is None or i - target < best[1] - best[0]):
            best = (target, i)
    if best is None:
        return None
    for _ in range(2):
        for i in range(best[0], best[1] + 1):
            step(i, code[i])
    return sorted(set(reports))


def main():
    unit, names = sys.argv[1], sys.argv[2:]
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--offload-device-only", "-S", unit, "-o", "-"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
    bad = 0
    for name, body in kernels(asm):
        if names and not any(n in name for n in names):
            continue
        rep = check(body)
        if rep is None:
            print("%-110s no window loop found" % name[:110])
            bad += 1
            continue
        print("%-110s %d load instructions, %d reports" % (name[:110], sum(1 for l in body if "global_load_dwordx4" in l), len(rep)))
        for i, text in rep[:12]:
            print("    line %5d: %s" % (i, text))
        bad += len(rep)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
