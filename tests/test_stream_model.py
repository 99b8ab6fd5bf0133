"""The stream kernel's control logic on the CPU (tools/sim/stream_model.py): a lane-level restatement of what a wave of
pire_amd/csrc/stream.hip does -- the cut of the batch by cost key, the window sequence, StepChunkB, StreamBoundary,
ExactRest, the boundary at the end of a line, the result slots -- on a toy automaton with few dense rows (frequent traps),
against a plain walk of every string.  Host only: the kernel itself is checked on the GPU against the oracle
(tests/test_gpu_parity.py::test_stream_kernel_vs_oracle); this catches the logic errors that would otherwise cost GPU
minutes -- that no line outside the text's own lines is ever fetched is asserted inside the model."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "sim"))
import stream_model as M  # noqa: E402


@pytest.mark.parametrize("kind,n,lead", [("urls", 1500, 0), ("urls", 300, 77), ("tiny", 2500, 5), ("lines", 300, 128),
                                         ("mixed", 1500, 1), ("aligned", 500, 0), ("aligned", 500, 112), ("edges", 1500, 3),
                                         ("empty", 1300, 9), ("empty", 200, 0), ("empty", 200, 128), ("one_long", 400, 0),
                                         ("urls", 64, 0), ("urls", 1025, 31)])
@pytest.mark.parametrize("seed", [1, 2])
def test_stream_model_equals_a_plain_walk(kind, n, lead, seed):
    fetched, lines = M.run_case(kind, n, lead, seed, base=4096 * 3, waves=6, min_units=64 * 64)
    assert fetched <= lines   # never a line outside the lines that hold the text (also asserted per fetch)


def test_stream_model_text_that_ends_with_a_page():
    """The last string ends exactly where a page ends and empty strings follow: nothing behind the page may be fetched
    (the model asserts every fetched line lies inside the text's lines)."""
    M.run_case("aligned", 257, 0, 5, base=4096 * 8, waves=3, min_units=64 * 16)
    M.run_case("empty", 100, 0, 5, base=4096 * 8, waves=3, min_units=64 * 16)
