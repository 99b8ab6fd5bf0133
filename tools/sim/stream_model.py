#!/usr/bin/env python3
"""Lane-level model of the stream kernel's control logic (pire_amd/csrc/stream.hip), run on the CPU against a plain
per-string walk.  No GPU is involved: the model restates, statement by statement and with the same 32-bit modular
arithmetic, what a wave of ScanStreamKernel does -- the cut of the batch by cost key (tasks, sub-tasks, lanes), the
window sequence with the virtual first phase, the per-chunk choice between the plain step and StepChunkB, boundaries
(StreamBoundary), the exact re-walk (ExactRest), the boundary at the end of a line, the result slots and the flush --
on a toy automaton whose dense rows cover only a few states, so that traps are frequent.  It exists because a logic
error in that kernel costs GPU minutes to find (round 4 lost 25 to one), and it is what tests/test_stream_model.py runs.

usage: stream_model.py [seed]"""
import sys

import numpy as np

M32 = 0xFFFFFFFF
INF = 0xFFFFFFFF
MAXS = 1280


def u32(x):
    return x & M32


class Table:
    """A random DFA over bytes; perm ids = state ids; states < hot have dense rows (entry >= hot -> trap id `hot`)."""

    def __init__(self, rng, states=40, hot=6, alphabet=7):
        self.states, self.hot = states, hot
        cls = rng.randint(0, alphabet, size=256)
        small = rng.randint(0, states, size=(states, alphabet))
        # make the hot states sticky so that walks spend most of their time in them, as ranked tables do
        for s in range(hot):
            for a in range(alphabet):
                if rng.rand() < 0.85:
                    small[s, a] = rng.randint(0, hot)
        self.nxt = small[:, cls]                      # [states][256]
        self.start = int(rng.randint(0, hot))         # the kernel requires a dense-row start state

    def hot_lookup(self, h, byte):
        if h >= self.hot:
            return self.hot                            # row H is absorbing
        n = int(self.nxt[h, byte])
        return n if n < self.hot else self.hot

    def slow_step(self, st, byte):
        return int(self.nxt[st, byte])


def reference(tab, text, offs):
    out = []
    for i in range(len(offs) - 1):
        st = tab.start
        for b in text[int(offs[i]):int(offs[i + 1])]:
            st = tab.slow_step(st, int(b))
        out.append(st)
    return np.array(out, dtype=np.int64)


class Lane:
    __slots__ = ("wpos", "E", "laneEnd", "dataEnd", "nxt", "sEnd", "hs", "cold", "live")


def run_model(tab, text, offs, text_base=0, lam=16, min_task_units=64 * 256, total_waves=8, check_reads=True):
    """offs: uint64 array [n+1]; text: the bytes at absolute addresses text_base + offset.  Returns end states."""
    n = len(offs) - 1
    out = np.full(n, -1, dtype=np.int64)
    off0, offN = int(offs[0]), int(offs[n])
    total_key = (offN - off0) + lam * n
    K = max(1, min(total_waves, total_key // min_task_units))
    per_task = (total_key + K - 1) // K
    key = (offs.astype(np.int64) - off0) + lam * np.arange(n + 1)
    mem_lo, mem_hi = text_base + off0, text_base + offN       # bytes of the batch
    line_lo = mem_lo & ~127
    line_hi = (mem_hi + 127) & ~127 if mem_hi > mem_lo else line_lo
    fetched = set()

    def lower_bound(T):                                         # StreamSearch2: smallest i in [0, n] with key(i) >= T
        lo, hi = 0, n
        while lo < hi:
            step = (hi - lo) // 64 + 1
            f = 64
            for lane in range(64):
                pos = lo + lane * step
                k = key[pos] if pos <= hi else 1 << 62
                if k >= T:
                    f = lane
                    break
            if f == 0:
                hi = lo
            else:
                last_below, first_at = lo + (f - 1) * step, lo + f * step
                lo = last_below + 1
                if f < 64 and first_at < hi:
                    hi = first_at
        return lo

    for gw in range(total_waves):
        if gw >= K:
            continue
        i0 = lower_bound(min(gw * per_task, total_key))
        i1 = lower_bound(min((gw + 1) * per_task, total_key))
        if gw == K - 1:
            i1 = n
        sub = i0
        task_strings = i1 - i0
        sub_tasks = (task_strings + MAXS - 1) // MAXS
        sub_strings = (task_strings + sub_tasks - 1) // sub_tasks if sub_tasks else 0
        while sub < i1:
            m = min(sub_strings, i1 - sub)
            offA, offZ = int(offs[sub]), int(offs[sub + m])
            first_byte = text_base + offA
            line_base = first_byte & ~127
            lead = first_byte & 127
            assert offZ - offA < 0xFFFF0000
            eo = [u32(lead + (int(offs[sub + q]) - offA)) for q in range(m + 1)]
            key_all = u32((offZ - offA) + lam * m)
            per_lane = (key_all + 63) // 64
            s0s = []
            for lane in range(64):
                target = u32(lane * per_lane)
                lo, hi = 0, m
                for _ in range(11):
                    mid = (lo + hi) >> 1
                    below = lo < hi and u32(u32(eo[mid] - lead) + lam * mid) < target
                    shrink = lo < hi and not below
                    lo = mid + 1 if below else lo
                    hi = mid if shrink else hi
                s0s.append(lo)
            lanes = []
            for lane in range(64):
                S = Lane()
                s0 = s0s[lane]
                s1 = m if lane == 63 else s0s[lane + 1]
                S.nxt, S.sEnd, S.live = s0, s1, False
                has = s0 < s1
                S.E = eo[s0] if has else INF
                S.laneEnd = eo[s1] if has else 0
                S.dataEnd = S.laneEnd if S.E < S.laneEnd else 0
                S.wpos = u32((S.E & ~127 & M32) - 128) if has else 0
                S.hs = S.cold = tab.start
                lanes.append(S)
            cur = [None] * 64                        # the line each lane holds (absolute address), None = dummy
            nxt_tile = [None] * 64

            def boundary(S, state):
                if S.live:
                    eo[S.nxt] = state
                if S.nxt < S.sEnd:
                    S.E = eo[S.nxt + 1]
                    S.nxt += 1
                    S.live = True
                else:
                    S.E = INF
                    S.live = False

            def chunk_bytes(lane, k):
                a = cur[lane]
                if a is None:
                    return [0xAA] * 16               # a dummy line: whatever
                return [int(text[a + 16 * k + j - text_base]) if mem_lo <= a + 16 * k + j < mem_hi else 0x55 for j in range(16)]

            def exact_rest(S, v, k, frm, st):
                for i in range(frm, 16):
                    while u32(S.E - S.wpos) == 16 * k + i:
                        boundary(S, st)
                        st = tab.start
                    if S.live:
                        st = tab.slow_step(st, v[i])
                S.hs = st if st < tab.hot else tab.hot
                S.cold = st

            walk = False
            guard = 0
            while True:
                guard += 1
                assert guard < 100000, "the window loop does not end"
                # ---- StreamPhase
                more = [S.laneEnd > u32(S.wpos + 128) for S in lanes]
                for lane, S in enumerate(lanes):
                    if S.dataEnd > u32(S.wpos + 128):
                        addr = line_base + u32(S.wpos + 128)
                        if check_reads:
                            assert line_lo <= addr and addr + 128 <= line_hi, ("line outside the text's lines", hex(addr))
                        fetched.add(addr)
                        nxt_tile[lane] = addr
                    else:
                        nxt_tile[lane] = None
                if walk:
                    for k in range(8):
                        cs = [u32(u32(S.E - S.wpos) - 16 * k) for S in lanes]
                        if not any(c < 16 for c in cs):
                            for lane, S in enumerate(lanes):
                                v = chunk_bytes(lane, k)
                                hs0 = S.hs
                                h = S.hs
                                for j in range(16):
                                    h = tab.hot_lookup(h, v[j])
                                S.hs = h
                                if h == tab.hot and S.live:          # TrapChunk
                                    st = hs0 if hs0 != tab.hot else S.cold
                                    for j in range(16):
                                        st = tab.slow_step(st, v[j])
                                    if st < tab.hot:
                                        S.hs = st
                                    else:
                                        S.hs, S.cold = tab.hot, st
                        else:
                            for lane, S in enumerate(lanes):
                                v = chunk_bytes(lane, k)
                                c = cs[lane]
                                hs0 = S.hs
                                h, sn = S.hs, S.hs
                                for j in range(16):                   # StepChunkB
                                    if c == j:
                                        sn = h
                                        h = tab.start
                                    h = tab.hot_lookup(h, v[j])
                                S.hs = h
                                is_b = c < 16
                                trap_before = S.live and ((sn == tab.hot) if is_b else (S.hs == tab.hot))
                                trap_after = is_b and S.hs == tab.hot and S.nxt < S.sEnd
                                exact = trap_before or trap_after
                                frm, st = 0, (hs0 if hs0 != tab.hot else S.cold)
                                if not exact and is_b:
                                    boundary(S, sn)
                                    if u32(u32(S.E - S.wpos) - 16 * k) < 16:
                                        exact, frm, st = True, c, tab.start
                                if exact:
                                    exact_rest(S, v, k, frm, st)
                for S in lanes:
                    while u32(S.E - S.wpos) == 128:
                        boundary(S, S.hs if S.hs != tab.hot else S.cold)
                        S.hs = S.cold = tab.start
                    S.wpos = u32(S.wpos + 128)
                cur, nxt_tile = nxt_tile, [None] * 64
                walk = True
                if not any(more):
                    break
            for S in lanes:
                assert S.E == INF and not S.live, "a lane did not finish its strings"
            for q in range(m):
                assert out[sub + q] == -1, "a string was finished twice"
                out[sub + q] = eo[q + 1]
            sub += sub_strings
    return out, fetched, (line_lo, line_hi)


CASES = {
    "urls": lambda rng, n: rng.randint(20, 200, size=n),
    "tiny": lambda rng, n: rng.randint(0, 9, size=n),
    "lines": lambda rng, n: rng.randint(64, 1024, size=n),
    "empty": lambda rng, n: np.zeros(n, dtype=np.int64),
    "aligned": lambda rng, n: rng.randint(0, 5, size=n) * 128 + rng.randint(0, 3, size=n) * 16,
    "edges": lambda rng, n: np.array([0, 1, 15, 16, 17, 31, 32, 112, 113, 127, 128, 129, 143, 144, 145, 255, 256, 257, 383, 384, 400])[
        rng.randint(0, 21, size=n)],
    "mixed": lambda rng, n: np.where(rng.rand(n) < 0.15, 0, np.where(rng.rand(n) < 0.1, rng.randint(100, 3000, size=n),
                                                                      rng.randint(0, 64, size=n))),
    "one_long": lambda rng, n: np.where(np.arange(n) == n // 3, 70000, rng.randint(0, 40, size=n)),
}


def run_case(kind, n, lead, seed, base=0, waves=8, min_units=64 * 256):
    rng = np.random.RandomState(seed)
    tab = Table(rng)
    ln = CASES[kind](rng, n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[0] = lead
    offs[1:] = lead + np.cumsum(ln)
    total = int(offs[-1])
    text = rng.randint(0, 256, size=max(total, 1)).astype(np.uint8)[:total]
    want = reference(tab, text, offs)
    got, fetched, (lo, hi) = run_model(tab, np.concatenate([np.zeros(0, np.uint8), text]), offs, text_base=base,
                                       total_waves=waves, min_task_units=min_units)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (kind, n, lead, seed, bad[:10], ln[bad[:10]], got[bad[:10]], want[bad[:10]])
    # every line of the text is read exactly where needed: none outside [lo, hi) (asserted inside)
    return len(fetched), (hi - lo) // 128


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for kind, n, lead in (("urls", 3000, 0), ("urls", 500, 77), ("tiny", 4000, 5), ("lines", 700, 128), ("mixed", 3000, 1),
                          ("aligned", 900, 0), ("aligned", 900, 112), ("edges", 3000, 3), ("empty", 2500, 9), ("empty", 300, 0),
                          ("empty", 300, 128), ("one_long", 600, 0), ("urls", 64, 0), ("urls", 1025, 31)):
        f, lines = run_case(kind, n, lead, seed, base=4096 * 3 + 0, waves=8, min_units=64 * 64)
        print("ok", kind, n, lead, "lines fetched", f, "of", lines)
