"""The two facts the segmented scan's mode handling rests on (segmented.hip ModeFunction / EnsureModeProduct), checked
on the host with the oracle alone -- no GPU, no product code: they are statements about automata.

1. Walk a text from the start state a0 and from a mode's representative b0 at once.  If the breadth-first search over
   the product automaton finds that, among the pairs reachable by texts of >= L bytes, every first component has one
   partner (b = f(a)), then for EVERY text of >= L bytes the second walk's state is f of the first's.
2. Whether or not such an f exists, one walk of the product automaton (pairs reachable from (a0, b0)) is the two walks."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

L = 256


def table_of(o):
    reps = {}
    for c in range(256):
        reps.setdefault(o.letter_class(c), c)
    chars = list(reps.values())
    nxt = np.zeros((o.size, len(chars)), dtype=np.int64)
    for s in range(o.size):
        for j, c in enumerate(chars):
            nxt[s, j] = o.next(s, c)
    col = np.zeros(256, dtype=np.int64)
    for c in range(256):
        col[c] = chars.index(reps[o.letter_class(c)])
    return nxt, col


def state_after(o, text, start=None):
    t = np.frombuffer(bytes(text), dtype=np.uint8)
    kw = {} if start is None else {"init_idx": np.array([start], dtype=np.uint32)}
    idx, _ = o.run(t, np.array([0, len(t)], dtype=np.uint64), flags=ob.FLAG_BEGIN if start is None else 0, **kw)
    return int(idx[0])


def mode_function(nxt, a0, b0):
    n = nxt.shape[0]
    level = np.array([[a0, b0]], dtype=np.int64)
    for _ in range(L):
        pr = np.unique(nxt[level[:, 0]].reshape(-1) * n + nxt[level[:, 1]].reshape(-1))
        new = np.stack([pr // n, pr % n], axis=1)
        if len(new) == len(level) and (new == level).all():
            break
        level = new
    f, todo = {}, []
    for a, b in level:
        if f.setdefault(int(a), int(b)) != int(b):
            return None
        todo.append(int(a))
    while todo:
        a = todo.pop()
        for j in range(nxt.shape[1]):
            na, nb = int(nxt[a, j]), int(nxt[f[a], j])
            if na not in f:
                f[na] = nb
                todo.append(na)
            elif f[na] != nb:
                return None
    return f


def texts(rng, alphabet, count, words):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    for _ in range(count):
        t = bytearray(a[rng.randint(0, len(a), size=rng.randint(L, 3 * L))].tobytes())
        if rng.randint(0, 2):
            w = words[rng.randint(0, len(words))]
            pos = rng.randint(0, len(t) - len(w))
            t[pos:pos + len(w)] = w
        yield bytes(t)


@pytest.mark.skipif(not ob.ref_available(), reason="needs oracle/_ref to compile the scanner")
def test_a_sticky_mode_of_unanchored_patterns_is_a_function_of_the_walk_from_the_start_state():
    o = ob.OracleScanner(ob.RefScanner.compile(["error", "time ?out", "fa+tal"]).save())
    nxt, _ = table_of(o)
    a0 = state_after(o, b"")
    rng = np.random.RandomState(3)
    for seen in (b"xx error yy", b"a timeout b fatal c", b" error faatal timeout "):
        b0 = state_after(o, seen)
        f = mode_function(nxt, a0, b0)
        assert f is not None, seen
        for t in texts(rng, b"abcdefghijklmnopqrstuvwxyz  .,", 40, [b" error ", b"timeout", b"faaatal"]):
            assert state_after(o, seen + t) == f[state_after(o, t)]


def test_dollar_anchored_patterns_have_no_such_function_but_their_product_is_small():
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    nxt, col = table_of(o)
    text = ob.corpus_fill(big["corpus"]["seed"], 0, 64, 4096, H.plants_for(big)).reshape(-1)
    a0 = state_after(o, b"")
    b0 = state_after(o, text[:len(text) // 2].tobytes())          # a state deep inside the text: a sticky mode
    assert b0 != a0
    assert mode_function(nxt, a0, b0) is None                      # the walk in the mode keeps what the other forgets
    # the product automaton from (a0, b0) over the byte letters, and one walk of it = the two walks
    index, pairs, pnext = {(a0, b0): 0}, [(a0, b0)], []
    i = 0
    while i < len(pairs):
        a, b = pairs[i]
        row = []
        for j in range(nxt.shape[1]):
            key = (int(nxt[a, j]), int(nxt[b, j]))
            if key not in index:
                index[key] = len(pairs)
                pairs.append(key)
            row.append(index[key])
        pnext.append(row)
        i += 1
    assert len(pairs) < 6 * o.size and len(pairs) <= 255           # (162 on this table: every state gets a dense row)
    rng = np.random.RandomState(4)
    for _ in range(20):
        lo = rng.randint(0, len(text) - 4096)
        piece = text[lo:lo + rng.randint(1, 4096)]
        p = 0
        for c in piece:
            p = pnext[p][col[c]]
        assert pairs[p] == (state_after(o, piece.tobytes()), state_after(o, piece.tobytes(), start=b0))
