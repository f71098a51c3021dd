"""Randomised parity stress: random protocol masks, thresholds (incl. auto), filters, noise levels, submit splits and
one to four submits in flight; every flush event of the GPU pipeline must equal the oracle's.
    python tests/stress_gpu.py <seed> <rounds>      (tests/test_gpu_parity.py runs a short fixed campaign)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from tfrec_amd import api, synth  # noqa: E402


def campaign(seed: int, rounds: int, verbose: bool = True) -> int:
    rng = np.random.default_rng(seed)
    bad = 0
    for rnd in range(rounds):
        bad += _round(rng, rnd, verbose)
    return bad


def _round(rng, rnd, verbose):
    if True:
        n_streams = int(rng.integers(1, 9))
        n_blocks = int(rng.integers(2, 40))
        types = int(rng.choice([0x01, 0x02, 0x04, 0x08, 0x20, 0x07, 0x2F, 0x2E, 0x21, 0x0F]))
        thresh = int(rng.choice([0, 150, 300, 500, 900, 2000]))
        wide = int(rng.integers(0, 2))
        noise = int(rng.choice([64, 256, 512, 1024, 3000]))
        seed = int(rng.integers(1, 1 << 30))
        iq = np.stack([synth.gen_stream(seed, s, n_blocks, int(rng.choice([0x1F, 0x1F, 0x11, 0x0E, 0x00])), noise)
                       for s in range(n_streams)])
        cuts = sorted(set([0, n_blocks] + [int(x) for x in rng.integers(1, n_blocks, size=int(rng.integers(0, 4)))]))
        mb = max(b - a for a, b in zip(cuts, cuts[1:]))
        # two rounds in three: the product library; the third: the experiments build (csrc/knobs.h) with a random pipeline layout
        exp = int(rng.integers(0, 3)) == 0
        if exp:
            os.environ["TFREC_AMD_DEEP"] = str(int(rng.integers(0, 2)))  # read when the context is created
        with api.Receiver(n_streams, types, thresh, wide, max_blocks=mb, all_flushes=True, max_events=400000, experiments=exp) as r:
            evs = []
            k = 0
            pend = 0
            for a, b in zip(cuts, cuts[1:]):
                r.submit(np.ascontiguousarray(iq[:, a * 65536:b * 65536]))
                pend += 1
                if pend == api.FIFO_DEPTH or rng.integers(0, 2):  # sometimes several submits in flight
                    keep = int(rng.integers(0, pend))              # ... and sometimes the younger ones stay in flight
                    while pend > keep:
                        evs.append(r.drain())
                        pend -= 1
            while pend:
                evs.append(r.drain())
                pend -= 1
            ev = np.concatenate(evs) if evs else np.zeros(0, api.EVENT_DTYPE)
            ok = True
            for s in range(n_streams):
                o = O.Oracle(types, thresh, wide)
                o.process(iq[s])
                want = sorted(o.events_full())
                got = sorted(api.event_tuples_full(ev, s))
                if got != want:
                    ok = False
            unc = r.atan_uncertain()
            if r.fm_stats()["host_mismatch"]:  # a discriminator decision of the exact slow path differs from this host's libm
                ok = False
        os.environ.pop("TFREC_AMD_DEEP", None)
        if verbose:
            print("round %d: streams %d blocks %d types %02x thresh %d wide %d noise %d cuts %s events %d unc %d -> %s" % (
            rnd, n_streams, n_blocks, types, thresh, wide, noise, cuts, len(ev), unc, "ok" if ok else "MISMATCH"), flush=True)
        return 0 if ok else 1


if __name__ == "__main__":
    n_bad = campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 10)
    print("mismatches:", n_bad)
    sys.exit(1 if n_bad else 0)
