#!/usr/bin/env python3
"""Mint tests/golden/inverted_sync.json from the REAL reference: a stream whose TFA_2-family bursts arrive with
inverted FSK polarity (I and Q swapped), so that tfa2_decoder::store_bit (tfa2.cpp:294-300) locks on the complemented
sync word and prints "Inverted SYNC" -- the one stdout line only a BITS-mode replay of the host adapter can reproduce
(SURVEY 8b).  TEST INFRASTRUCTURE; runs only where /root/reference exists.

Usage: python oracle/mint_inverted.py
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mint_golden as M  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tfrec_amd import synth  # noqa: E402

CASES = [
    dict(seed=21, stream=0, n_blocks=24, proto_mask=0x0E, noise_q8=256, types=0x2F, thresh=500, wide=0, iq_swap=1),
    dict(seed=21, stream=1, n_blocks=24, proto_mask=0x1F, noise_q8=256, types=0x2F, thresh=500, wide=0, iq_swap=1),
]


def make_iq(c):
    iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
    if c.get("iq_swap"):
        iq = iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy()  # Q, I: the spectrum mirrored, every FSK deviation negated
    return iq


def main():
    O.build()
    assert O.have_reference(), "needs /root/reference (oracle/_ref/ref_driver)"
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for c in CASES:
            iq = make_iq(c)
            ref = M.compare_stream(iq, c["types"], c["thresh"], c["wide"], tmp, "inverted case %r" % c)
            n_inv = ref["text"].splitlines().count("Inverted SYNC")
            M.check(n_inv >= 3, "expected Inverted SYNC lines, got %d" % n_inv)
            d = dict(c)
            d.update(iq_sha256=M.sha(iq), events=M.events_to_json(ref["events"]), data=M.data_to_json(ref["data"]),
                     text=ref["text"], bits=ref["bits"])
            out.append(d)
            print("case stream=%d: %d flushes, %d text lines, %d x Inverted SYNC" % (
                c["stream"], len(ref["events"]), len(ref["text"].splitlines()), n_inv))
    with open(os.path.join(M.GOLD, "inverted_sync.json"), "w") as f:
        json.dump(dict(source="oracle/_ref/ref_driver run (real reference hot path) on tfrec_amd.synth streams with I and Q "
                              "swapped (oracle/mint_inverted.py)", cases=out), f, indent=0)


if __name__ == "__main__":
    main()
