"""The first-use self-test of the kernels behind pire_hip_run / pire_hip_run_strided (pire_hip_config.selftest, api.cpp SelfTest):
the first time a table takes one of them, that kernel scans a known-answer batch -- a walk through the table's own states --
and the library compares it with the host image's transitions; a mismatch returns PIRE_HIP_ESELFTEST and writes nothing.

The failure path is reached by altering the expected answer (selftest = 2); the library's own kernels pass (selftest = 0,
the default: every other GPU test of this suite runs with it)."""
import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests.test_gpu_parity import dev_run_strided, pa, torch_cuda  # noqa: F401  (fixtures)
from tests.test_wide import dev_run_offsets

ESELFTEST = -6


def _strided(torch, t, data):
    return dev_run_strided(torch, t, torch.as_tensor(data, device="cuda"))


def _blob(name):
    from tests import helpers as H

    return H.load_blob([b for b in H.big_sets() if b["name"] == name][0]["blob"])


KINDS = [
    # kernel, config, scanner, batch: ("strided", strings, length) | ("offsets", strings)
    ("tiled", dict(), "set_a", ("strided", 256, 1024)),
    ("wide", dict(walk_variant=2), "dict_1k", ("strided", 256, 1024)),
    ("ragged", dict(ragged_variant=1), "set_a", ("offsets", 3000)),
    ("stream", dict(ragged_variant=2), "set_a", ("offsets", 3000)),
    ("ragged_wide", dict(walk_variant=2), "dict_1k", ("offsets", 3000)),
    ("stream_wide", dict(walk_variant=2, ragged_variant=2), "dict_1k", ("offsets", 3000)),
    ("generic", dict(), "set_a", ("strided", 8, 100)),
]


def _scanner(name):
    if name.startswith("dict"):
        return W.load_blob(W.wide_set(name)["blob"])
    return _blob(name)


def _batch(kind, seed):
    rng = np.random.RandomState(seed)
    if kind[0] == "strided":
        return rng.randint(32, 127, size=(kind[1], kind[2])).astype(np.uint8), None
    lens = rng.randint(0, 200, size=kind[1])
    offs = np.zeros(kind[1] + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return rng.randint(32, 127, size=int(offs[-1])).astype(np.uint8), offs


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,config,scanner,kind", KINDS, ids=[k[0] for k in KINDS])
def test_first_use_runs_the_known_answer_batch_and_a_wrong_answer_refuses_the_call(pa, torch_cuda, cfg, kernel, config, scanner, kind):
    from pire_amd import binding as pb

    torch = torch_cuda
    blob = _scanner(scanner)
    o = ob.OracleScanner(blob)
    data, offs = _batch(kind, 11)

    def call(t):
        if offs is None:
            gi, gf, _ = _strided(torch, t, data)
            oi, of = o.run(data.reshape(-1), np.arange(data.shape[0] + 1, dtype=np.uint64) * data.shape[1], threads=2)
        else:
            gi, gf, _ = dev_run_offsets(torch, t, data, offs)
            oi, of = o.run(data, offs, threads=2)
        return (gi == oi).all() and (gf == of).all()

    # 1. the expected answer altered: the call is refused, with the kernel's name in the message -- and keeps being refused
    cfg.set(selftest=2, **config)
    t = pa.Table(blob)
    for _ in range(2):
        with pytest.raises(pb.PireHipError) as err:
            call(t)
        assert err.value.code == ESELFTEST
        assert f"self-test of the {kernel} kernel failed" in str(err.value), str(err.value)
        assert pb.build_info().split(";")[0] in str(err.value)
    # 2. the same table with the real expectation: passes once, results as the oracle's
    cfg.set(selftest=0, **config)
    assert call(t)
    assert pb.last_kernel() == kernel
    # 3. ... and is not asked again: with the fault switched back on the table's kernel stays trusted
    cfg.set(selftest=2, **config)
    assert call(t)
    # 4. switched off, a fresh table is never tested
    cfg.set(selftest=1, **config)
    assert call(pa.Table(blob))


@pytest.mark.gpu
def test_self_test_leaves_the_visit_counters_alone(pa, torch_cuda, cfg):
    """The batch walks states the caller's text may never see: its samples must not reach the ranking."""
    torch = torch_cuda
    t = pa.Table(_blob("set_b"))                  # thousands of states: the self-test's text leaves the dense rows
    data, _ = _batch(("strided", 64, 256), 3)
    data[:] = ord("a")
    _strided(torch, t, data)                      # first use: self-test + a scan that stays in very few states
    assert t.adapt() == 0
    info = t.info
    # the scan above is 64 x 256 steps in one state's row, too few for a sample; the self-test's 131 072 steps through
    # the whole table would have left dozens, in the dense rows' counters and in the trap counters
    assert info.adaptations == 0 and info.last_trap_samples == 0    # "never ran: nothing observed"


# ---- the entry points with actions (round 6, csrc/selftest.h): prefix / suffix / half-final searches, counting and capturing ----

def _golden_blob(name):
    from tests import helpers as H

    return H.load_blob([c for c in H.all_cases() if c["name"] == name][0]["blob"])


def _strings(seed, n=300):
    from tests import helpers as H

    return H.random_strings(np.random.RandomState(seed), n, 150, b"abc def,hello w0123456789()-XYZ@\n")


def _entry_calls(pa):
    """(label in the failure message, a callable making a FRESH table and one call of the entry point, kernels the self-test must reach)"""
    from tests import helpers as H

    set_a = _blob("set_a")
    strings = _strings(5)
    text, offs = H.pack(strings)
    counting = [c for c in H.golden()["counting"] if c["kind"] == 1 and c["regexps"] >= 2][0]
    capturing = H.golden()["capturing"][0]
    return [
        ("Prefix (kernel", lambda: pa.Table(set_a).prefix(text, offs, True), {"ragged_prefix", "prefix", "ragged_prefix_wide"}),
        ("Suffix (kernel", lambda: pa.Table(set_a).suffix(text, offs, True), {"suffix"}),
        ("HalfFinalScanner", lambda: pa.Table(set_a).run_half_final(text, offs), {"ragged_half_final", "half_final", "ragged_half_final_wide"}),
        ("the counting scanner", lambda: pa.CountingTable(H.load_blob(counting["blob"]), counting["kind"]).run_strings(strings),
         {"counting", "counting_packed", "counting_rows"}),
        ("the capturing scanner", lambda: pa.CountingTable(H.load_blob(capturing["blob"]), 0).capture(text, offs),
         {"capture", "capture_dense", "capture_rows"}),
    ]


@pytest.mark.gpu
def test_entry_points_with_actions_test_every_kernel_they_can_take_on_first_use(pa, torch_cuda, cfg):
    """selftest = 2 (one expected answer altered): the first call of each entry point on a fresh table is refused with
    PIRE_HIP_ESELFTEST and the entry point's name; selftest = 0: it passes, and the kernels the library reports as self-tested
    include every kernel the entry point can route to."""
    from pire_amd import binding as pb

    for label, call, kernels in _entry_calls(pa):
        cfg.set(selftest=2)
        with pytest.raises(pb.PireHipError) as err:
            call()
        assert err.value.code == ESELFTEST and "self-test of " in str(err.value) and label in str(err.value), (label, str(err.value))
        cfg.set(selftest=0)
        call()
        tested = set(pb.selftested_kernels())
        assert kernels <= tested, (label, kernels - tested, tested)
        cfg.set(selftest=1)
        call()   # switched off: nothing tested, nothing refused


def test_every_kernel_name_the_library_can_emit_has_a_self_test():
    """Every NoteKernel("...") name in the sources is either reached by a first-use self-test (the names the GPU test above and the
    KINDS table check) or listed here with the reason it is not."""
    import os
    import re

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pire_amd", "csrc")
    names = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp")):
            with open(os.path.join(csrc, f)) as fh:
                src = fh.read()
            for m in re.finditer(r"NoteKernel\(([^;]*?)\);", src, re.S):
                names.update(re.findall(r'"([a-z_+0-9]+)"', m.group(1).split(",")[0] if "?" not in m.group(1) else m.group(1).split(', "pirehip')[0]))
    covered = {k[0] for k in KINDS} | {"ragged_prefix", "ragged_prefix_wide", "prefix", "suffix", "ragged_half_final", "ragged_half_final_wide",
                                       "half_final", "half_final_rows", "counting",
                                       "counting_packed", "counting_rows", "counting_letter_rows", "capture", "capture_dense", "capture_rows",
                                       "ragged_capture"}
    not_covered = {
        "pair_tiled": "ScannerPair on fixed-length records: the fused pass is held against two plain passes by tests/test_pair.py; no first-use test yet",
        "segmented": "the segmented scan verifies itself: every segment's guessed start state is checked against the true one on the device",
        "segmented+plain": "as 'segmented'",
        "tiled_seg": "a pass of the segmented scan (kPermIds): see 'segmented'",
        "slow": "SlowScanner: no first-use test yet (plain HIP, no hand-counted waits)",
        "slow_list": "as 'slow'",
        "slow_wide": "as 'slow'",
    }
    assert names, "no NoteKernel calls found"
    missing = names - covered - set(not_covered)
    assert not missing, f"kernels without a first-use self-test and without a stated reason: {sorted(missing)}"
