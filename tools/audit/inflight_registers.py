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
        # (the loads issued from INLINE ASM: the block-wide table copy in front of the window loop is a loop of eight dwordx4 loads
        # too -- the compiler's own --, and round 6 found the walker looking at that one in the kernels where it is the shorter)
        n = sum(1 for j in range(target, i) if code[j].startswith("global_load_dwordx4") and from_asm[j])
        if n >= 8 and (best is None or i - target < best[1] - best[0]):
            best = (target, i)
    if best is None:
        return None
    for _ in range(2):
        for i in range(best[0], best[1] + 1):
            step(i, code[i])
    return sorted(set(reports))


def check_exits(body, outermost=True):
    """Round 6 (lessons 24 and 29): what happens to the tile registers on the ways OUT of the window loop.  A load
    issued from inline asm in the loop's last trip is still on its way when the loop is left; behind the loop the registers are
    somebody else's, and whatever names one of them before an `s_waitcnt vmcnt(0)` reads data that has not arrived or is
    overwritten when it does.  For every exit edge of the window loop (a branch out of it, the fall-through behind its back
    edge) the code is followed -- both arms of every branch, up to 4 000 instructions -- until it waits for all loads
    (s_waitcnt vmcnt(0)) or ends; an instruction that names a register any asm load of the loop writes, met before that, is
    reported.  [(line number, text)], or None when the kernel has no window loop.
    WHICH loop: by default the OUTERMOST back edge around the asm loads -- behind it lies the kernel's epilogue, where the two
    faults of round 6 were (the ragged kernel's counter flush; hipcc had moved its first instruction in front of a wait that was
    there).  outermost=False: the innermost one -- the ways out of a tile loop INSIDE a task loop, where round 5's ScanWideKernel
    waited behind the loop; in layout order that also follows blocks of the outer loop that the compiler laid out behind the
    inner back edge and that name tiles waited for long ago, so the build asks for it only where it is known to be quiet
    (build_audit.py INNER_EXITS)."""
    code, labels, from_asm = [], {}, []
    in_asm = False
    has_markers = any("#ASMSTART" in line for line in body)
    for line in body:
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        text = line.split(";")[0].strip()
        code.append(text)
        from_asm.append(in_asm or not has_markers)
        m = re.match(r"^(\.LBB\S+):", text)
        if m:
            labels[m.group(1)] = len(code) - 1
    best = None
    for i, text in enumerate(code):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", text)
        if not m:
            continue
        target = labels.get(m.group(1) or m.group(2))
        if target is None or target > i:
            continue
        # (the loads issued from INLINE ASM: the block-wide table copy in front of the window loop is a loop of eight dwordx4 loads
        # too -- the compiler's own --, and round 6 found the walker looking at that one in the kernels where it is the shorter)
        n = sum(1 for j in range(target, i) if code[j].startswith("global_load_dwordx4") and from_asm[j])
        if n >= 8 and (best is None or ((i - target > best[1] - best[0]) if outermost else (i - target < best[1] - best[0]))):
            best = (target, i)
    if best is None:
        return None
    lo, hi = best
    tiles = set()
    for i in range(lo, hi + 1):
        if code[i].startswith("global_load_dwordx4") and from_asm[i]:
            tiles |= regs(code[i].split()[1].rstrip(","))
    # exit edges
    starts = set()
    if not code[hi].startswith("s_branch"):
        starts.add(hi + 1)
    for i in range(lo, hi + 1):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", code[i])
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and (t > hi or t < lo):
                starts.add(t)
    reports, seen = [], set()
    stack = [(s0, 0, 0) for s0 in sorted(starts)]   # (instruction, depth, loads issued since the loop was left)
    while stack:
        i, depth, younger = stack.pop()
        while i < len(code) and depth < 4000:
            if (i, min(younger, 64)) in seen:
                break
            seen.add((i, min(younger, 64)))
            text = code[i]
            depth += 1
            if not text or text.endswith(":") or text.startswith("."):
                i += 1
                continue
            op = text.split()[0]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m and int(m.group(1)) <= younger:
                    break   # loads return in order: all but the `younger` loads issued since have landed, the loop's among them
                i += 1
                continue
            if lo <= i <= hi:
                break   # back inside the window loop (an outer loop's next trip): the loop's own waits take over
            if op.startswith("global_load") or op.startswith("scratch_load") or op.startswith("buffer_load") or \
               op.startswith("global_store") or op.startswith("scratch_store") or op.startswith("global_atomic"):
                if regs(" ".join(text.split()[2:])) & tiles and not (op.startswith("global_load_dwordx4") and from_asm[i]):
                    reports.append((i, text))
                    break
                younger += 1   # (an asm load that lands in a tile register again is the next task's request: queued behind the loop's)
                i += 1
                continue
            if regs(text) & tiles:
                reports.append((i, text))
                break
            m = re.match(r"^s_cbranch_\w+\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", text)
            if m:
                t = labels.get(m.group(1) or m.group(2))
                if t is not None:
                    stack.append((t, depth, younger))
                if op == "s_branch":
                    break
            i += 1
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
        ex = check_exits(body) or []
        print("%-110s on the ways out of the (outermost) window loop: %d reports" % ("", len(ex)))
        for i, text in ex[:12]:
            print("    line %5d: %s" % (i, text))
        bad += len(ex)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
