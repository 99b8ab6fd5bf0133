"""order.hip: the one-string-per-lane kernels with per-byte actions (counting scanners, the capture walks) take the
strings of a large offset batch ordered by length class, so that the 64 lanes of a wave finish together.  Results must
not depend on it: every string's answer against the oracle with the order on (the default from 32 768 strings) and off
(pire_hip_config.no_length_order), for every length class -- empty strings, every 32-byte step below 1 KiB, the quarter
octaves above, strings beyond the packed kernel's 65 000 bytes."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu


def batch(rng, alphabet, n=40000):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    lens = np.concatenate([rng.randint(0, 1100, size=n - 600), rng.randint(1024, 9000, size=560),
                           rng.randint(9000, 40000, size=37), np.array([0, 0, 70001])])
    rng.shuffle(lens)
    return [a[rng.randint(0, len(a), size=int(k))].tobytes() for k in lens]


@pytest.mark.parametrize("name", ["count_glued3_advanced", "count0_basic", "count0_noglue"])
def test_counting_kernels_do_not_depend_on_the_order(name, cfg):
    import pire_amd
    from pire_amd import binding as pb

    case = [c for c in H.golden().get("counting", []) if c["name"] == name]
    if not case:
        pytest.skip("fixture not present")
    blob = H.load_blob(case[0]["blob"])
    t, o = pire_amd.CountingTable(blob, case[0]["kind"]), ob.OracleCountingScanner(blob, case[0]["kind"])
    many = batch(np.random.RandomState(7), b"abc def,http:/\n")
    oi, orr = o.run_strings(many)
    for off in (0, 1):
        cfg.set(no_length_order=off)
        for flags in (3, 3 | pb.FLAG_GENERIC):
            gi, gr = t.run_strings(many, flags=flags)
            assert (gi == oi).all() and (gr == orr).all(), (name, off, flags, pb.last_kernel())
    assert orr.sum() > 0


@pytest.mark.parametrize("name", ["capture_digits", "capture_path"])
def test_capture_kernels_do_not_depend_on_the_order(name, cfg):
    import pire_amd
    from pire_amd import binding as pb

    case = [c for c in H.golden().get("capturing", []) if c["name"] == name]
    if not case:
        pytest.skip("fixture not present")
    blob = H.load_blob(case[0]["blob"])
    t, o = pire_amd.CountingTable(blob, 0), ob.OracleCountingScanner(blob, 0)
    many = batch(np.random.RandomState(8), b"google_id ='\";x1/=0123456789to-match-with")
    many = [s[:len(s) // 3] + b" /to-match-with=42;" + s[len(s) // 3:] if i % 50 == 0 else s for i, s in enumerate(many)]
    want = o.capture(*ob.pack_strings(many))
    cfg.set(no_ragged_act=1)                      # the one-string-per-lane kernels (dense rows, or letter + transition)
    for off in (0, 1):
        cfg.set(capture_by_length=1 - off)        # (off by default for these kernels: measured slower)
        for flags in (3, 3 | pb.FLAG_GENERIC):
            got = t.capture(*H.pack(many), flags=flags)
            assert pb.last_kernel() in ("capture_dense", "capture")
            assert all((x == y).all() for x, y in zip(want, got)), (name, off, flags)
    assert want[2].sum() > 0


@pytest.mark.parametrize("name", ["slow_x40_utf8", "slow_alt"])
def test_slow_list_kernel_does_not_depend_on_the_order(name, cfg):
    """The SlowScanner list kernel takes ragged batches by length class too (it is all VALU: a wave waiting for its
    longest string is its one avoidable cost)."""
    import pire_amd
    from pire_amd import binding as pb

    case = [c for c in H.golden().get("slow", []) if c["name"] == name]
    if not case:
        pytest.skip("fixture not present")
    blob = H.load_blob(case[0]["blob"])
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    rng = np.random.RandomState(12)
    many = [s for s in batch(rng, b"ax.yd e\xd0\xb6bcx")]
    of, obits = o.run_strings(many)
    for off in (0, 1):
        cfg.set(no_length_order=off)
        gf, gbits, cnt = t.run_strings(many, counts=True)
        assert pb.last_kernel().startswith("slow")
        assert (gf == of).all() and (gbits == obits).all(), (name, off)
        assert cnt.tolist() == [int(of.sum()), len(many)]
