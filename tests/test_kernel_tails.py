"""Every kernel that keeps text on its way in registers, hammered where its END is short (round 6, DESIGN.md 6 lesson 29): a wave
whose last iteration has nothing to walk leaves its loop with loads still on their way, and what the epilogue then does with the
registers is only right if something waited.  The ragged kernel had that window since round 2 and nothing ever hit it -- a
fault once in a few hundred launches; found by tools/stress_dict.py.  Here: each kernel kind on inputs that end it as early as
they can -- records that all reach the absorbing state in their first tile (the wave-wide early-out), batches that end in runs
of empty strings, strings that die at their first byte -- a few hundred launches each, every answer the oracle's."""
import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests import helpers as H
from tests.test_gpu_parity import dev_run_strided, pa, torch_cuda  # noqa: F401  (fixtures)
from tests.test_wide import dev_run_offsets, records_of

pytestmark = pytest.mark.gpu
REPS = 300


def _dict():
    entry = W.wide_set("dict_1k")
    return entry, W.load_blob(entry["blob"])


@pytest.mark.parametrize("walk,zipv", [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2)], ids=["dense rows", "wide", "wide x2", "zipped", "zipped x2"])
@pytest.mark.parametrize("n,length", [(64 * 40, 256), (64 * 40 + 17, 4096), (1 << 15, 384)])
def test_fixed_length_kernels_where_every_record_is_absorbed_at_once(pa, torch_cuda, cfg, walk, zipv, n, length):
    """A word of the dictionary at byte 3 of EVERY record: all lanes of every wave are in the absorbing state after the first
    tile, every task ends by the wave-wide early-out with its next tiles requested."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry, blob = _dict()
    cfg.set(walk_variant=walk, zip_variant=zipv, auto_adapt=1)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    data = records_of(entry, "k128", 99, n, length).copy()
    word = np.frombuffer(W.dictionary_words(entry)[11], dtype=np.uint8)
    data[:, 3:3 + len(word)] = word
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    assert of.all()
    d = torch.as_tensor(data, device="cuda")
    for r in range(REPS):
        gi, gf, _ = dev_run_strided(torch, t, d)
        assert (gi == oi).all() and (gf == of).all(), (r, pb.last_kernel_symbol())
    assert pb.last_kernel() == ("tiled" if walk == 1 else "wide")


@pytest.mark.parametrize("walk,zipv,raggedv", [(1, 1, 1), (1, 1, 2), (2, 1, 1), (2, 2, 1), (2, 1, 2), (2, 2, 2)],
                         ids=["ragged", "stream", "ragged wide", "ragged zipped", "stream wide", "stream zipped"])
def test_offset_batch_kernels_on_batches_that_end_in_nothing(pa, torch_cuda, cfg, walk, zipv, raggedv):
    """Some text, then thousands of empty strings; only empty strings; strings of one byte."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry, blob = _dict()
    cfg.set(walk_variant=walk, zip_variant=zipv, ragged_variant=raggedv, auto_adapt=1, no_offsets_peek=1)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.upload()
    text = records_of(entry, "k128", 5, 64, 1024).reshape(-1)
    cases = []
    for lens in (np.r_[np.full(300, 200), np.zeros(5000)], np.zeros(7000), np.ones(9000), np.r_[np.zeros(4000), [3000], np.zeros(4000)]):
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens.astype(np.uint64))
        tx = text[:max(int(offs[-1]), 1)]
        cases.append((tx, offs) + o.run(tx[:int(offs[-1])], offs, threads=2))
    for r in range(REPS // 2):
        for tx, offs, oi, of in cases:
            gi, gf, _ = dev_run_offsets(torch, t, tx, offs)
            assert (gi == oi).all() and (gf == of).all(), (r, len(offs), pb.last_kernel_symbol())


@pytest.mark.parametrize("walk", [1, 2], ids=["dense rows", "wide rows"])
def test_walks_with_actions_whose_strings_are_over_at_once(pa, torch_cuda, cfg, walk):
    """A blacklist scanner walked without BeginMark: every string Dead at its first byte, every search over before its first
    chunk ends (the launch that found the fault); and the half-final counting on empty strings."""
    from pire_amd import binding as pb

    entry = W.wide_set("blacklist_1k")
    blob = W.load_blob(entry["blob"])
    cfg.set(walk_variant=walk, zip_variant=0, auto_adapt=1, ragged_act_always=1, no_offsets_peek=1)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    text, offs = W.wide_urls(entry, 3, 20000)
    want = {lg: o.prefix(text, offs, lg, False, False) for lg in (True, False)}
    assert (want[True] < 0).all()
    empty = np.zeros(6001, dtype=np.uint64)
    hi, hf, hr = o.run_half_final(text[:0], empty)
    for r in range(REPS):
        for lg in (True, False):
            assert (t.prefix(text, offs, lg, False, False) == want[lg]).all(), (r, lg)
        gi, gf, gr = t.run_half_final(np.zeros(16, dtype=np.uint8), empty)
        assert (gi == hi).all() and (gf == hf).all() and (gr == hr).all(), r
