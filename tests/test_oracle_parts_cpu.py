"""oracle.process_parts (receivers that continue over several batches: what the steady-state GPU test and bench.py's
parity-after-the-timed-region check compare against) equals the oracle over the concatenated bytes -- the reference
reads one long dump block by block (engine.cpp:63-93), so cutting a stream into batches must change nothing."""
import numpy as np

from oracle import oracle as O
from tfrec_amd import synth


def test_parts_equal_one_long_stream_and_keep_from_selects_the_tail():
    a = synth.gen_batch(11, 0, 3, 7)
    b = synth.gen_batch(12, 0, 3, 5)
    whole = O.process_many(np.concatenate([a, b, b], axis=1), 0x2F, 500, 0)
    parts = O.process_parts([a, b], 0x2F, 500, 0, reps=[1, 2])
    assert sum(len(x) for x in whole) > 30
    for s in range(3):
        assert np.array_equal(whole[s], parts[s])
    # only the events of the last repetition (end_sample still counted from the stream's first sample)
    last = O.process_parts([a, b, b], 0x2F, 500, 0, keep_from=2)
    for s in range(3):
        w = whole[s][whole[s]["end_sample"] >= 12 * 8192]
        assert np.array_equal(w, last[s])
    # a column slice of a longer array is a valid part (rows not adjacent): no copy is needed
    long_iq = np.concatenate([a, b], axis=1)
    views = [long_iq[:, :7 * 65536], long_iq[:, 7 * 65536:]]
    sliced = O.process_parts(views, 0x2F, 500, 0)
    both = O.process_many(long_iq, 0x2F, 500, 0)
    for s in range(3):
        assert np.array_equal(sliced[s], both[s])
