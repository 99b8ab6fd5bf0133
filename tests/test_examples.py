"""examples/pigrep_hip.cpp: the reference's sample grep (samples/pigrep) with the scan on the GPU must print exactly
what the reference's own pigrep prints.  Both binaries are built here (where /root/reference exists) against the
unmodified reference headers and travel to the GPU box prebuilt; the inputs are this repository's own documents."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "bin")
REF_PRESENT = os.path.exists("/root/reference/samples/pigrep/pigrep.cpp")

CASES = [
    (["-i", "lds.*bytes"], ["DESIGN.md"]),
    (["v_perm"], ["DESIGN.md", "README.md", "INTEGRATION.md"]),
    (["-x", "(kernel&~ragged)"], ["DESIGN.md"]),
    (["-u", "-i", "§[0-9]"], ["DESIGN.md", "SURVEY.md"]),
    (["-e", "-m gpu"], ["README.md", "DESIGN.md"]),
    (["^$"], ["DESIGN.md"]),                       # empty lines
    (["no such text anywhere 12345"], ["DESIGN.md"]),
    (["[0-9]+\\.[0-9]+ (TB|GB)/s"], ["DESIGN.md", "BASELINE.md", "README.md"]),
]


def run(binary, args, files, stdin=None):
    r = subprocess.run([os.path.join(BIN, binary)] + args + files, cwd=ROOT, input=stdin, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not present (GPU box): the prebuilt binaries are used there")
def test_examples_build_against_reference_headers():
    from oracle import binding as ob

    ob.build()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert os.path.exists(os.path.join(BIN, "pigrep_hip")) and os.path.exists(os.path.join(BIN, "pigrep_ref"))


@pytest.mark.gpu
@pytest.mark.parametrize("args,files", CASES)
def test_pigrep_hip_prints_what_pigrep_prints(args, files):
    if not os.path.exists(os.path.join(BIN, "pigrep_hip")):
        pytest.skip("examples/bin was not built (needs /root/reference at build time)")
    want = run("pigrep_ref", args, files)
    got = run("pigrep_hip", args, files)
    assert got[0] == 0, got[2][-2000:]
    assert got[1] == want[1]
    if "no such" not in args[-1]:
        assert want[1], "case selects nothing: it does not test anything"


@pytest.mark.gpu
def test_pigrep_hip_stdin_and_unterminated_last_line():
    if not os.path.exists(os.path.join(BIN, "pigrep_hip")):
        pytest.skip("examples/bin was not built (needs /root/reference at build time)")
    text = b"alpha\n\nbeta gamma\n" + b"x" * 5000 + b" gamma\nlast gamma without newline"
    for args in (["gamma"], ["^$"], ["-i", "ALPHA|x{4000}"]):
        want = run("pigrep_ref", args, [], stdin=text)
        got = run("pigrep_hip", args, [], stdin=text)
        assert got[0] == 0, got[2][-2000:]
        assert got[1] == want[1] and want[1]
    assert run("pigrep_hip", ["gamma"], [], stdin=b"")[1] == b""


def test_pigrep_hip_fails_loudly_without_gpu():
    """Not a GPU test: on a box without a GPU the example must fail with a message, not fall back to a CPU scan."""
    import torch

    if torch.cuda.is_available() or not os.path.exists(os.path.join(BIN, "pigrep_hip")):
        pytest.skip("needs the built example and no GPU")
    rc, out, err = run("pigrep_hip", ["kernel"], ["DESIGN.md"])
    assert rc != 0 and out == b"" and b"pigrep_hip:" in err


@pytest.mark.gpu
def test_pigrep_hip_on_a_file_with_very_long_lines(tmp_path):
    """Lines of a megabyte and more next to short ones: the batch goes through the segmented scan; same output."""
    if not os.path.exists(os.path.join(BIN, "pigrep_hip")):
        pytest.skip("examples/bin was not built (needs /root/reference at build time)")
    import random

    rnd = random.Random(7)
    words = [b"alpha", b"beta", b"gamma delta", b"0123456789", b"   ", b"needle-42", b"x"]
    lines = []
    for k in range(6):
        parts, size = [], 0
        target = (1 << 20) + 1000 * k if k % 2 == 0 else rnd.randint(10, 300)
        while size < target:
            w = words[rnd.randrange(len(words) - 1)] if k != 4 else words[rnd.randrange(len(words))]
            parts.append(w)
            size += len(w)
        lines.append(b" ".join(parts))
    path = tmp_path / "long_lines.txt"
    path.write_bytes(b"\n".join(lines) + b"\n")
    for args in (["needle-[0-9]+"], ["-i", "GAMMA\\s+delta$"], ["^alpha"]):
        want = run("pigrep_ref", args, [str(path)])
        got = run("pigrep_hip", args, [str(path)])
        assert got[0] == 0, got[2][-2000:]
        assert got[1] == want[1]
