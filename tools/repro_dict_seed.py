"""Repeats the prefix / half-final calls of one seed of tests/test_random_dictionaries.py (a fault that shows once in a few runs)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
import tests.test_random_dictionaries as T
from tests.test_gpu_parity import stream_lengths

seed = int(sys.argv[1]); zipv = int(sys.argv[2]); reps = int(sys.argv[3]); adapt = int(sys.argv[4]); which = sys.argv[5]
rng = np.random.RandomState(7000 + seed)
words, symbols, mode = T.draw_dictionary(rng)
blob = ob.RefScanner.compile_dictionary(words, surround=(mode == 1), utf8=(mode == 2)).save()
o = ob.OracleScanner(blob)
n, length = 2048, int(rng.choice([384, 1024, 1152]))
data = T.draw_text(rng, words, symbols, n * length).reshape(n, length)
kind = ["urls", "tiny", "mixed", "lines", "edges", "aligned"][int(rng.randint(0, 6))]
m = int(rng.choice([300, 5000, 20000]))
ln = stream_lengths(rng, kind, m).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64); offs[1:] = np.cumsum(ln)
text = T.draw_text(rng, words, symbols, max(int(offs[-1]), 1))[:int(offs[-1])]
print(seed, len(words), mode, o.size, o.letters, kind, m, flush=True)
pb.set_config(zip_variant=zipv, auto_adapt=1, no_offsets_peek=1, ragged_act_always=1, walk_variant=int(os.environ.get('WALK', '2')))
t = pire_amd.Table(blob)
d = torch.as_tensor(data, device="cuda")
for _ in range(adapt):
    t.run_strided_host(data); t.run(text, offs); t.adapt()
print("info", t.refresh_info().wide_states, t.refresh_info().zip_full_states, flush=True)
want = {(lg, tb): o.prefix(text, offs, lg, tb, tb) for lg in (True, False) for tb in (True, False)}
for r in range(reps):
    for (lg, tb), w in want.items():
        if which == "all" or which == f"{int(lg)}{int(tb)}":
            got = t.prefix(text, offs, lg, tb, tb)
            assert (got == w).all(), (r, lg, tb)
    if which in ("all", "hf"):
        t.run_half_final(text, offs)
print("ok", reps, pb.last_kernel(), flush=True)
