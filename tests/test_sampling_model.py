"""The arithmetic of the visit samples (round 6, DESIGN.md 3.1 and lesson 30), restated on the CPU: WHICH lane, byte and step the
kernels draw.  Rounds 2-5 drew them from counters that every wave, block and launch starts at 0 -- the same handful of draws
everywhere, for ever: a state every string is in at its first byte was never looked at.  The constants are the kernels'
(ragged.hip RaggedPhase / ScanRaggedKernel, wide_common.h WideTrapChunk); if they change there, change them here."""
import numpy as np

PHI, SEED = 0x9E3779B1, 0x632BE5AB
M32 = 0xFFFFFFFF


def ragged_draws(blocks, waves_per_block, iterations, seeded=True):
    """{(lane, byte)} drawn by a launch: sampleHash = iter * PHI with iter starting at the wave's seed (or at 0: rounds 2-5)."""
    seen = set()
    for b in range(blocks):
        for w in range(waves_per_block):
            it0 = ((b * 16 + w) * SEED) & M32 if seeded else 0
            it = (it0 + np.arange(iterations, dtype=np.uint64)) & M32
            h = (it * PHI) & M32
            seen.update(zip((h >> 26).tolist(), ((h >> 19) & 127).tolist()))
    return seen


def test_every_lane_and_byte_of_a_window_is_drawn_within_one_launch_of_a_url_batch():
    # a URL batch: 256 blocks x 16 waves, 32 iterations each -- 131 072 draws over 64 x 128 = 8 192 pairs
    seen = ragged_draws(256, 16, 32)
    assert len(seen) == 64 * 128
    # ... and byte 0 -- the state every string starts in stands in front of it -- by every lane
    assert {lane for lane, byte in seen if byte == 0} == set(range(64))
    # rounds 2-5: every wave the same 32 draws
    old = ragged_draws(256, 16, 32, seeded=False)
    assert len(old) == 32 and not any(byte == 0 for _, byte in old if _ != 0)


def rewalk_steps(blocks, rewalks_per_block, seeded=True):
    """The steps (0..15) whose front state the sampled re-walks of a launch leave: every 64th re-walk of a block, the step drawn
    from the re-walk's number (and the block's)."""
    steps = set()
    for b in range(blocks):
        for nth in range(0, rewalks_per_block, 64):
            steps.add(((((nth >> 6) + (b * SEED if seeded else 0)) & M32) * PHI & M32) >> 28)
    return steps


def test_the_re_walks_sample_leaves_every_step_of_a_chunk_within_one_launch():
    # a few hundred re-walks per block and launch (a URL batch): every step of the sixteen
    assert rewalk_steps(256, 320) == set(range(16))
    # drawn from the block's own count alone: the first five draws, the same in every block and launch
    assert rewalk_steps(256, 320, seeded=False) == {0, 9, 3, 13, 7}
