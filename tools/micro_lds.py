#!/usr/bin/env python3
"""The LDS ceiling of the dense-row walk for a table and the corpus it is benchmarked on (tools/micro/micro_lds.hip).

    python tools/micro_lds.py [set_d] [--waves 8] [--steps 1024]

Builds the trace -- the LDS address (dense id << 8 | byte) of every step of 64 consecutive strings of the synthetic corpus,
per wave -- from the table's HOST accessors (pire_hip_table_next / _letter_class / _layout) and the corpus generator, and
runs the replay on the GPU.  Reports what the LDS sustains for exactly this pattern of addresses next to what the product
kernel gets on the same table (bench.py --set <name>)."""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import pire_amd
from oracle import binding as ob   # the corpus generator's host twin (test infrastructure: this is a measurement tool)
from pire_amd import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("set", nargs="?", default="set_d")
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out")
    args = ap.parse_args()
    big = W.pattern_set(args.set)
    t = pire_amd.Table(W.load_blob(big["blob"]))
    info = t.info
    orig_of_perm, hot_rows = t.layout()
    hot = info.hot_states
    # next state by byte for the dense rows only (the walk of the corpus stays inside them: bench.py `traps`)
    n, length = args.waves * 64, 4096
    text = ob.corpus_fill(0x5EED5EED, 0, n, length, ob.make_plants([(bytes.fromhex(h), tl) for h, tl in zip(big["witnesses_hex"], big["witness_at_tail"])]), threads=4)
    perm_of_orig = np.empty_like(orig_of_perm)
    perm_of_orig[orig_of_perm] = np.arange(len(orig_of_perm), dtype=np.uint32)
    # the exact table by byte (host accessors; device numbering), so that a lane that leaves the dense rows comes back as it
    # does in the kernel (its lookups meanwhile go to the trap row, id == hot)
    cls = np.array([t.letter_class(b) for b in range(256)])
    rep = {int(c): b for b, c in reversed(list(enumerate(cls)))}
    nxt = np.empty((info.states, 256), dtype=np.int64)
    for o in range(info.states):
        by_cls = {c: int(perm_of_orig[t.Next(o, b)]) for c, b in rep.items()}
        nxt[perm_of_orig[o]] = [by_cls[int(c)] for c in cls]
    st = np.full(n, int(perm_of_orig[t.Next(info.initial, 258)]), dtype=np.int64)   # Begin()
    start = 1024   # walk to a position well inside the strings, then record `steps` steps
    for i in range(start):
        st = nxt[st, text[:, i]]
    addr = np.empty((n, args.steps), dtype=np.uint16)
    left = 0
    for i in range(args.steps):
        b = text[:, start + i].astype(np.int64)
        addr[:, i] = (np.minimum(st, hot) << 8) | b
        left += int((st >= hot).sum())
        st = nxt[st, b]
    states_per_step = float(np.mean([len(np.unique(addr[w * 64:(w + 1) * 64, i] >> 8)) for w in range(args.waves) for i in range(0, args.steps, 16)]))
    # [waves][steps / 16][64 lanes][16]
    tr = addr.reshape(args.waves, 64, args.steps // 16, 16).transpose(0, 2, 1, 3).copy()
    os.makedirs(args.out, exist_ok=True)
    tpath, rpath = os.path.join(args.out, "micro_trace.bin"), os.path.join(args.out, "micro_rows.bin")
    tr.tofile(tpath)
    hot_rows.tofile(rpath)
    print("%s: %d states, %d dense rows; %d strings, steps %d..%d of each; %.1f distinct dense rows per wave-step; %d lane-steps outside the dense rows"
          % (args.set, info.states, hot, n, start, start + args.steps, states_per_step, left))
    exe = os.path.join(HERE, "micro", "micro_lds")
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", os.path.join(HERE, "micro", "micro_lds.hip"), "-o", exe], check=True)
    sys.stdout.flush()
    return subprocess.run([exe, tpath, rpath, str(args.waves), str(args.steps), str(args.reps)]).returncode


if __name__ == "__main__":
    sys.exit(main())
