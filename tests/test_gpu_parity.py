"""GPU parity tests: the HIP path (through the C ABI) against the pinned CPU oracle, bit-exact.

Integer / byte / index work => equality, no tolerance.  The only floating point on the path (fp64
discriminator + biquads) feeds integer truncations; those integers are compared exactly too.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tfrec_amd import api, synth

pytestmark = pytest.mark.gpu


def oracle_events(iq, types, thresh, wide=0):
    o = O.Oracle(types, thresh, wide, keep_dec=True)
    o.process(iq)
    return o


def by_slot(evs):
    d = {}
    for e in evs:
        d.setdefault(e[0], []).append(e)
    return d


def check_stream(gpu_events, stream, orc, min_bytes_only=False):
    # full tuples: ..., rssi_raw (the accumulator itself, not only its dB value: BASELINE.md 3 "raw RSSI and offset integers
    # identical"), status (the decoder's CRC / sanity verdict of every event, computed on the GPU)
    g = by_slot(api.event_tuples_full(gpu_events, stream))
    oe = orc.events_full()
    if min_bytes_only:
        minb = {0: 10, 1: 7, 2: 7, 3: 7, 4: 11}
        oe = [e for e in oe if e[2] >= minb[e[0]] and not (e[0] == 3 and e[2] >= 64) and not (e[0] == 4 and e[2] > 60)]
    o = by_slot(oe)
    assert sorted(g.keys()) == sorted(o.keys())
    for slot in o:
        assert g[slot] == o[slot], "stream %d slot %d" % (stream, slot)
    return sum(len(v) for v in o.values())


@pytest.mark.parametrize("wide", [0, 1])
def test_frontend_bit_exact(wide):
    n_streams, n_blocks = 6, 3
    iq = synth.gen_batch(3, 100, n_streams, n_blocks)
    iq[1] = np.random.default_rng(1).integers(0, 256, iq.shape[1], dtype=np.uint8)  # full-scale random bytes
    iq[2, :] = 0
    iq[3, :] = 255
    with api.Receiver(n_streams, 0x2F, 500, wide, max_blocks=n_blocks) as r:
        r.submit(iq)
        for s in range(n_streams):
            want = np.empty(2 * n_blocks * 8192, dtype=np.int16)
            O.lib().orc_decimate(iq[s].ctypes.data, iq.shape[1] // 2, wide, want.ctypes.data)
            got = r.decimated(s, n_blocks * 8192)
            assert np.array_equal(got, want), "stream %d" % s


SERIAL = pytest.mark.parametrize("serial", [False, True], ids=["pipeline", "serial"])


@SERIAL
def test_all_flush_events_match_oracle(serial):
    n_streams, n_blocks = 24, 48
    iq = synth.gen_batch(7, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, serial_chains=serial) as r:
        r.submit(iq)
        ev = r.drain()
        total = 0
        for s in range(n_streams):
            total += check_stream(ev, s, oracle_events(iq[s], 0x2F, 500))
        assert total > 20 * n_streams
        assert r.fm_stats()["host_mismatch"] == 0


def test_default_mode_reports_candidates_with_verdict():
    n_streams, n_blocks = 8, 48
    iq = synth.gen_batch(9, 50, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks) as r:
        r.submit(iq)
        ev = r.drain()
        for s in range(n_streams):
            orc = oracle_events(iq[s], 0x2F, 500)
            check_stream(ev, s, orc, min_bytes_only=True)
            # status==1 events are exactly the flushes that produced telegram text in the reference
            n_ok = int(np.sum((ev["stream"] == s) & (ev["status"] == 1)))
            lines = [ln for ln in orc.text().splitlines() if not ln.startswith("Inverted") and not ln.startswith("WHB:")]
            assert n_ok == len(lines)


@SERIAL
@pytest.mark.parametrize("all_flushes", [True, False], ids=["all_flushes", "default"])
def test_planted_crc_and_sanity_failures_status_per_event(serial, all_flushes):
    """Every other burst carries a planted fault (synth.gen_stream corrupt_every: wrong checksum / right checksum but a field
    the decoder's sanity test rejects / frame cut short, in turn): the verdict the GPU computes for EVERY event (status:
    tfa1.cpp:63-73, tfa2.cpp:93/237, whb.cpp:506-510 + the length tests) equals the oracle's, event by event, and so do the
    raw RSSI accumulators."""
    n_streams, n_blocks = 10, 48
    iq = np.stack([synth.gen_stream(31, s, n_blocks, corrupt_every=2) for s in range(n_streams)])
    seen = {0: 0, 1: 0, 2: 0}
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=all_flushes, serial_chains=serial) as r:
        r.submit(iq)
        ev = r.drain()
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], 0x2F, 500), min_bytes_only=not all_flushes)
        for st in seen:
            seen[st] = int(np.sum(ev["status"] == st))
    # planted: rejected telegrams of every protocol next to accepted ones
    rej = ev[ev["status"] == 2]
    assert seen[1] >= 5 * n_streams and seen[2] >= 5 * n_streams and sorted(set(rej["slot"].tolist())) == [0, 1, 2, 3, 4]
    assert (seen[0] > 0) == all_flushes


@SERIAL
def test_state_carries_across_submits(serial):
    n_streams = 5
    iq = synth.gen_batch(11, 7, n_streams, 24)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=16, all_flushes=True, serial_chains=serial) as r:
        evs = []
        pos = 0
        for nb in (1, 7, 3, 13):
            r.submit(np.ascontiguousarray(iq[:, pos * 65536:(pos + nb) * 65536]))
            evs.append(r.drain())
            pos += nb
        ev = np.concatenate(evs)
        ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], 0x2F, 500))


@SERIAL
@pytest.mark.parametrize("types,thresh", [(0x01, 500), (0x07, 500), (0x20, 300), (0x0E, 1500)])
def test_type_masks_and_thresholds(types, thresh, serial):
    n_streams, n_blocks = 6, 24
    iq = synth.gen_batch(13, 3, n_streams, n_blocks, noise_q8=512)
    with api.Receiver(n_streams, types, thresh, 0, max_blocks=n_blocks, all_flushes=True, serial_chains=serial) as r:
        r.submit(iq)
        ev = r.drain()
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], types, thresh))


def test_device_resident_input_and_timings():
    import torch

    n_streams, n_blocks = 16, 8
    iq = synth.gen_batch(17, 0, n_streams, n_blocks)
    d = torch.from_numpy(iq).cuda()
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, timing=True) as r:
        r.submit(d)
        ev = r.drain()
        t = r.timings()
        assert t["frontend_ms"] > 0 and t["chains_ms"] > 0
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], 0x2F, 500))


def _stress_batch():
    """Inputs that open and close trigger windows as often as possible (noise hovering around the threshold),
    saturate them (uniform random bytes), or pack bursts tightly: exercises window tables at their limits,
    the tfa2 last_bit_idx speculation repair, biquad convergence failures and block-boundary corner cases."""
    rng = np.random.default_rng(5)
    n_blocks = 12
    rows = []
    for k, (noise, mask) in enumerate([(2048, 0x1F), (1536, 0x1F), (1024, 0x1F), (768, 0x0E), (3072, 0x00), (2560, 0x11)]):
        rows.append(synth.gen_stream(77, k, n_blocks, mask, noise))
    rows.append(rng.integers(0, 256, n_blocks * 65536, dtype=np.uint8))           # always triggered
    r = rng.integers(0, 256, n_blocks * 65536, dtype=np.uint8)
    gate = (np.arange(r.size) // 3000) % 2 == 0                                     # on/off every 1500 samples
    rows.append(np.where(gate, r, 128).astype(np.uint8))
    gate2 = (np.arange(r.size) // 1700) % 3 == 0
    rows.append(np.where(gate2, r, (128 + rng.integers(-2, 3, r.size))).astype(np.uint8))
    return np.stack(rows)


@SERIAL
@pytest.mark.parametrize("thresh", [350, 500, 900])
def test_stress_sporadic_triggers(serial, thresh):
    iq = _stress_batch()
    n_streams, n_blocks = iq.shape[0], iq.shape[1] // 65536
    with api.Receiver(n_streams, 0x2F, thresh, 0, max_blocks=n_blocks, all_flushes=True, serial_chains=serial,
                      max_events=400000) as r:
        # two submits so that windows straddle the submit boundary as well
        r.submit(np.ascontiguousarray(iq[:, : 5 * 65536]))
        e1 = r.drain()
        r.submit(np.ascontiguousarray(iq[:, 5 * 65536:]))
        e2 = r.drain()
        ev = np.concatenate([e1, e2])
        ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
        total = 0
        for s in range(n_streams):
            total += check_stream(ev, s, oracle_events(iq[s], 0x2F, thresh))
        assert total > 500


@SERIAL
def test_auto_threshold_matches_oracle(serial):
    """-t 0: the reference's adaptive trigger threshold (fm_demod.cpp:23-27, 58-73), block by block."""
    n_blocks = 48
    rows = [synth.gen_stream(7, 3, n_blocks, 0x1F, 512),    # the golden auto-threshold case of streams.json
            synth.gen_stream(19, 1, n_blocks, 0x1F, 1536),   # noisier: threshold climbs
            synth.gen_stream(19, 2, n_blocks, 0x1F, 64),     # quiet: threshold falls to its floor region
            synth.gen_stream(19, 3, n_blocks, 0x00, 2048)]
    iq = np.stack(rows)
    with api.Receiver(iq.shape[0], 0x2F, 0, 0, max_blocks=20, all_flushes=True, serial_chains=serial,
                      max_events=200000) as r:
        evs = []
        for a, b in ((0, 20), (20, 33), (33, 48)):  # thresholds must carry across submits
            r.submit(np.ascontiguousarray(iq[:, a * 65536:b * 65536]))
            evs.append(r.drain())
        ev = np.concatenate(evs)
        ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
        moved = 0
        for s in range(iq.shape[0]):
            o = oracle_events(iq[s], 0x2F, 0)
            check_stream(ev, s, o)
            assert r.thresh(s) == o.thresh(), "stream %d" % s
            moved += int(o.thresh() != 500)
        assert moved >= 2


def test_submits_in_flight_fifo():
    """Submits k+1 .. k+3 may be queued before submit k is drained (FIFO of depth four); a fifth one is refused."""
    n_streams = 6
    iq = synth.gen_batch(23, 5, n_streams, 40)
    parts = [np.ascontiguousarray(iq[:, a * 65536:b * 65536]) for a, b in ((0, 8), (8, 16), (16, 24), (24, 32), (32, 40))]
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=10, all_flushes=True) as r:
        ref = []
        for p in parts:
            r.submit(p)
            ref.append(r.drain())
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=10, all_flushes=True) as r:
        import torch
        dev = [torch.from_numpy(p).cuda() for p in parts]
        got = []
        assert api.FIFO_DEPTH == 4
        for k in range(4):
            r.submit(dev[k])
        with pytest.raises(RuntimeError):
            r.submit(dev[4])  # four submits are waiting to be drained
        got.append(r.drain())
        r.submit(dev[4])
        for k in range(4):
            got.append(r.drain())
        assert len(r.drain()) == 0
    for a, b in zip(ref, got):
        assert len(a) == len(b) and len(a) > 0
        assert a.tobytes() == b.tobytes()


def test_drain_fetches_events_beyond_the_copy_queued_at_submit(monkeypatch):
    """The submit queues the device-to-host copy of a GUESSED number of events (twice the last drain's count); a batch
    with more events than that must still deliver all of them, in the same order."""
    n_streams = 6
    iq = synth.gen_batch(29, 11, n_streams, 24)
    parts = [np.ascontiguousarray(iq[:, a * 65536:b * 65536]) for a, b in ((0, 2), (2, 14), (14, 24))]
    out = {}
    for guess in (None, "1"):
        if guess:
            monkeypatch.setenv("TFREC_AMD_COPY_GUESS_MIN", guess)
        with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=12, all_flushes=True, experiments=True) as r:
            for p in parts:  # queued together: the guesses of the 2nd and 3rd submit come from before the 1st drain
                r.submit(p)
            out[guess] = [r.drain() for _ in parts]
    assert sum(len(e) for e in out[None]) > 50
    for a, b in zip(out[None], out["1"]):
        assert a.tobytes() == b.tobytes()
    ev = np.concatenate(out["1"])
    for s in range(n_streams):
        check_stream(ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))], s, oracle_events(iq[s], 0x2F, 500))


def test_config5_10x_front_end():
    """BASELINE config 5: 15.36 MS/s u8 input -> 10:1 integer FIR (defined by this project, oracle.decim10) ->
    the standard path on int16 input.  The 10:1 stage is checked bit for bit against its C restatement, everything
    after it against the oracle's int16 entry (which the real reference pins through ref_driver run16)."""
    n_streams, n_blocks = 3, 12
    iq = np.stack([synth.gen_stream(31, s, n_blocks, 0x1F, 256, rate_mult=10) for s in range(n_streams)])
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=8, all_flushes=True, input_10x=True) as r:
        evs = []
        for a, b in ((0, 5), (5, 12)):  # history of both FIR stages carries across submits
            r.submit(np.ascontiguousarray(iq[:, a * 655360:b * 655360]))
            evs.append(r.drain())
            for s in range(n_streams):
                want = O.decim10(iq[s, : b * 655360])[2 * a * 32768:]
                got = r.stage0(s, (b - a) * 32768)
                assert np.array_equal(got, want), "10:1 stage, stream %d" % s
        ev = np.concatenate(evs)
        ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
        total = 0
        for s in range(n_streams):
            o = O.Oracle(0x2F, 500, 0)
            o.process_s16(O.decim10(iq[s]))
            total += check_stream(ev, s, o)
        assert total >= 10


def test_config5_full_length_batch_of_64_streams():
    """BASELINE config 5 at the benchmark's submit length (48 blocks, 31.5 M input samples per stream) on a batch of 64 streams --
    16 distinct ones, each four times: the 10:1 stage against its C restatement on every distinct stream, the events of all 64
    against the oracle's int16 entry on the 10:1 stage's output, and replicas identical wherever they sit in the batch."""
    n_unique, copies, n_blocks = 16, 4, 48
    base = np.stack([synth.gen_stream(53, s, n_blocks, 0x1F, 256, rate_mult=10) for s in range(n_unique)])
    iq = np.concatenate([base] * copies)
    n_streams = iq.shape[0]
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, input_10x=True, max_events=n_streams * 400) as r:
        r.submit(iq)
        ev = r.drain()
        assert r.fm_stats()["host_mismatch"] == 0
        stage0 = [r.stage0(s, n_blocks * 32768) for s in range(n_unique)]
    ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
    total = 0
    for s in range(n_unique):
        want = O.decim10(base[s])
        assert np.array_equal(stage0[s], want), "10:1 stage, stream %d" % s
        o = O.Oracle(0x2F, 500, 0)
        o.process_s16(want)
        for k in range(copies):
            total += check_stream(ev, s + k * n_unique, o)
    assert total >= 40 * copies


def test_full_size_pipeline_equals_serial_chains_and_replicas():
    """BASELINE-size streams (48 blocks): the window-parallel pipeline and the independent one-lane-per-chain GPU
    implementation must agree on every flush event, and identical input streams must give identical events
    wherever they sit in the batch (size-independent properties; the oracle itself is checked on a few streams)."""
    n_unique, copies, n_blocks = 48, 4, 48
    base = synth.gen_batch(41, 100, n_unique, n_blocks, noise_q8=384)
    iq = np.concatenate([base] * copies)  # stream s and s + k*n_unique carry the same samples
    n_streams = iq.shape[0]

    def run(serial):
        with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, serial_chains=serial,
                          max_events=n_streams * 400) as r:
            r.submit(iq)
            ev = r.drain()
            assert r.fm_stats()["host_mismatch"] == 0
            return ev

    a, b = run(False), run(True)
    assert len(a) == len(b) and len(a) > 20 * n_streams
    assert a.tobytes() == b.tobytes()
    first = a[a["stream"] < n_unique]
    for k in range(1, copies):
        rep = a[(a["stream"] >= k * n_unique) & (a["stream"] < (k + 1) * n_unique)].copy()
        rep["stream"] -= k * n_unique
        assert rep.tobytes() == first.tobytes()
    for s in (0, 17, 47):
        check_stream(a, s, oracle_events(base[s], 0x2F, 500))


def test_tfa2_edge_timing_speculation_failure_is_resliced_exactly():
    """A TFA_2-family window is sliced before its predecessor's last_bit_idx is known (assumed far in the past); when
    the assumption fails the commit step re-slices it with the exact value.  These two synthetic streams hit that path
    on their second pass (found by profiles/ubench/find_reslice.py): the re-slice must have happened, and the events
    must still be the oracle's."""
    n_blocks = 12
    iq = np.concatenate([synth.gen_batch(1000, s, 1, n_blocks) for s in (2387, 3079)])
    with api.Receiver(2, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True) as r:
        evs = []
        for _ in range(2):
            r.submit(iq)
            evs.append(r.drain())
        assert r.stats()["tfa2_resliced"] > 0
        ev = np.concatenate(evs)
        ev = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
    for s in range(2):
        check_stream(ev, s, oracle_events(np.concatenate([iq[s], iq[s]]), 0x2F, 500))


def test_deep_and_shallow_layouts_agree_across_submits(monkeypatch):
    """Deep layout (six streams: the biquad stage of submit k+1 beside the slicers of submit k, two table sets) and
    shallow layout must give the same events as each other and as the oracle over a sequence of unequal submits kept
    as deep in the FIFO as it goes."""
    n_streams, cuts = 6, (3, 1, 5, 2, 4, 1)
    iq = synth.gen_batch(77, 40, n_streams, sum(cuts), noise_q8=320)
    got = {}
    for deep in ("1", "0"):
        monkeypatch.setenv("TFREC_AMD_DEEP", deep)
        with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=max(cuts), all_flushes=True, experiments=True) as r:
            assert r.layout() == (6 if deep == "1" else 4)
            evs, pos, pending = [], 0, 0
            for nb in cuts:
                r.submit(np.ascontiguousarray(iq[:, pos * 65536:(pos + nb) * 65536]))
                pos += nb
                pending += 1
                if pending == api.FIFO_DEPTH:
                    evs.append(r.drain())
                    pending -= 1
            while pending:
                evs.append(r.drain())
                pending -= 1
            ev = np.concatenate(evs)
            got[deep] = ev[np.lexsort((ev["seq"], ev["slot"], ev["stream"]))]
    assert got["1"].tobytes() == got["0"].tobytes()
    for s in range(n_streams):
        check_stream(got["1"], s, oracle_events(iq[s], 0x2F, 500))


def test_randomised_campaign():
    """Random masks / thresholds (incl. auto) / filters / noise / submit splits / submits in flight (tests/stress_gpu.py)."""
    import stress_gpu
    assert stress_gpu.campaign(3, 32, verbose=False) == 0


def test_fm_dev_next_to_truncation_boundaries_equals_the_real_reference(golden_dir):
    """dsp_stuff.cpp:284-292 on inputs whose scaled angle is within 1e-9 of an integer: the device's exact slow path
    against goldens of the REAL reference (oracle/mint_fm_boundary.py) -- and every logged decision against this
    host's libm (the drain-time self check)."""
    import os

    g = np.load(os.path.join(golden_dir, "fm_boundary.npz"))
    got, st = api.fm_dev_probe(g["quads"])
    assert np.array_equal(got, g["quads_ref"])
    assert st["resolved"] > 5000 and st["host_verified"] > 50 and st["host_mismatch"] == 0
    # directions as close as 1e-19 rad to a boundary (cross terms below 2^31): the result under a correctly rounded atan2
    got, st = api.fm_dev_probe(g["cross"], cross=True)
    assert np.array_equal(got, g["cross_rn"])
    assert st["undecidable"] > 1000  # the fixture holds 1074 vectors inside glibc's 0.55-ulp band: counted, never silent
    # random + octant / axis probes of the unit fixture
    u = np.load(os.path.join(golden_dir, "unit_probes.npz"))
    got, st = api.fm_dev_probe(u["fm_in"])
    assert np.array_equal(got, u["fm_out"][:, 0]) and st["host_mismatch"] == 0


@pytest.mark.parametrize("form", [0, 1])
def test_iir2_probe_equals_the_real_reference_bit_for_bit(golden_dir, form):
    """iir2::set / iir2::step (dsp_stuff.cpp:28-56) by themselves: the device's biquad -- as the reference associates the step
    (form 0) and in the 3-multiply form every biquad pass and WHB stage 2 run (form 1) -- over the 4000-sample probe sequence,
    for the five cut-offs the reference instantiates (main.cpp:186-217, tfa2.cpp:321, whb.cpp:610-611), against outputs of the
    REAL reference (oracle/mint_golden.py: unit_probes.npz).  The pipeline tests see the biquads only through the integers
    behind them; this one localises a failure."""
    import os

    u = np.load(os.path.join(golden_dir, "unit_probes.npz"))
    for k, cutoff in enumerate(u["iir_cutoffs"]):
        y = api.iir_probe(float(cutoff), u["iir_in"], form=form)
        assert np.array_equal(y.view(np.uint64), u["iir_out"][k].view(np.uint64)), "cutoff %r form %d" % (cutoff, form)


@pytest.mark.parametrize("eps", ["1e-4", "1e-3", "0.6"])
def test_fm_dev_slow_path_through_the_pipeline(eps, monkeypatch):
    """The exact slow path decides the same integer as the fast path wherever the fast path is certain, so widening
    the flag threshold (TFREC_AMD_FM_FLAG_EPS) drives it -- the deferred list (1e-4: ~40 entries per submit), its
    overflow (1e-3: more than the list holds) and the whole-submit rescan (0.6: every sample) -- through the real pipeline with
    ordinary input: events stay equal to the oracle's, and every logged decision equals this host's libm."""
    monkeypatch.setenv("TFREC_AMD_FM_FLAG_EPS", eps)
    n_streams, n_blocks = 4, 16
    iq = synth.gen_batch(31, 7, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, experiments=True) as r:
        for h in range(2):  # two submits: the carried state after a patched submit
            r.submit(np.ascontiguousarray(iq[:, h * (n_blocks // 2) * 65536:(h + 1) * (n_blocks // 2) * 65536]))
        ev = np.concatenate([r.drain(), r.drain()])
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], 0x2F, 500))
        st = r.fm_stats()
        assert st["resolved"] > 30 and st["host_verified"] > 30 and st["host_mismatch"] == 0
        assert st["undecidable"] == 0


@pytest.mark.parametrize("splits", [1, 3])
def test_bits_mode_every_store_bit_equals_the_oracle(splits):
    """SURVEY 8b 'kind = BITS': with TFREC_AMD_F_BITS every bit the demodulators hand to decoder::store_bit (tfa1.cpp:120,
    tfa2.cpp:281 incl. the 16 trailing bits, whb.cpp:566 incl. the 16 zeros before a flush) comes back, grouped by the
    flush it precedes; compared flush by flush with the oracle's bit log (= the real reference's 'W' records,
    tests/test_oracle_golden.py) for all five slots, also when the stream is cut into several submits."""
    n_streams, n_blocks = 5, 24
    iq = synth.gen_batch(43, 3, n_streams, n_blocks)
    cut = n_blocks // splits
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=cut, all_flushes=True, bits=True, max_events=1 << 17) as r:
        evs = []
        for k in range(splits):
            r.submit(np.ascontiguousarray(iq[:, k * cut * 65536:(k + 1) * cut * 65536]))
            evs.append(r.drain())
        ev = np.concatenate(evs)
    n_bits = 0
    for s in range(n_streams):
        o = O.Oracle(0x2F, 500, 0, log_bits=True)
        o.process(iq[s])
        want = {}
        for ln in o.bits_text().splitlines():  # "W slot nbits bits": one record per flush, in flush order
            p = ln.split()
            want.setdefault(int(p[1]), []).append(p[3] if len(p) > 3 else "")
        got = api.bits_by_flush(ev, s)
        check_stream(ev, s, o)  # the flush events themselves are unchanged by the mode
        for slot, recs in want.items():
            for seq, bits in enumerate(recs):
                assert got.get((slot, seq), "") == bits, "stream %d slot %d flush %d" % (s, slot, seq)
                n_bits += len(bits)
        assert sorted(want.keys()) == [0, 1, 2, 3, 4]
    assert n_bits > 20000


@pytest.mark.parametrize("vec", ["1", "0"])
def test_cooperative_slicers_step_per_lane_and_scalar_walk_emit_the_same_bits(vec, monkeypatch):
    """Long windows (tfa1.cpp:150-178, tfa2.cpp:383-411) are sliced by a wave per window: 64 steps at a time with a step per lane
    (default), or by the scalar walk (TFREC_AMD_TFA1_VEC=0 / TFREC_AMD_TFA2_VEC=0: what the lane-per-step form falls back to for a
    group it gives up).  Both must hand decoder::store_bit the oracle's bits, flush by flush -- also across submits that cut
    the windows (carried last_bit_idx / mark_lvl / thresholds) and through noise above the threshold (one window per stream)."""
    monkeypatch.setenv("TFREC_AMD_TFA1_VEC", vec)
    monkeypatch.setenv("TFREC_AMD_TFA2_VEC", vec)
    n_streams, n_blocks, cut = 6, 36, 12
    iq = synth.gen_batch(47, 11, n_streams, n_blocks)
    noisy = synth.gen_stream(47, 99, n_blocks, 0x1F, 16 * 256)  # noise sigma 16 LSB: above -t 500, the trigger never drops
    iq = np.concatenate([iq, noisy[None, :]])
    n_streams += 1
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=cut, all_flushes=True, bits=True, max_events=1 << 18, experiments=True) as r:
        evs = []
        for k in range(n_blocks // cut):
            r.submit(np.ascontiguousarray(iq[:, k * cut * 65536:(k + 1) * cut * 65536]))
            evs.append(r.drain())
        ev = np.concatenate(evs)
        st = r.stats()
    if vec == "0":
        # (the counters describe the lane-per-step form: nothing to count when it is switched off)
        assert st["tfa1_scalar_groups"] == 0 and st["tfa2_scalar_groups"] == 0
        assert st["tfa1_vector_groups"] == 0 and st["tfa2_vector_groups"] == 0
    else:
        # the lane-per-step form RAN (a regression that silently routes everything to the scalar walks would pass the bit
        # comparison below), did most groups, and its fallback was exercised too (the noisy stream: every sample a rejected
        # candidate, more than 16 rounds of re-walking)
        assert st["tfa1_vector_groups"] > 0 and st["tfa2_vector_groups"] > 0, st
        assert st["tfa1_vector_groups"] > st["tfa1_scalar_groups"] and st["tfa2_vector_groups"] > st["tfa2_scalar_groups"], st
        assert st["tfa1_scalar_groups"] + st["tfa2_scalar_groups"] > 0, st
    n_bits = 0
    for s in range(n_streams):
        o = O.Oracle(0x2F, 500, 0, log_bits=True)
        o.process(iq[s])
        want = {}
        for ln in o.bits_text().splitlines():
            p = ln.split()
            want.setdefault(int(p[1]), []).append(p[3] if len(p) > 3 else "")
        got = api.bits_by_flush(ev, s)
        check_stream(ev, s, o)
        for slot, recs in want.items():
            for seq, bits in enumerate(recs):
                assert got.get((slot, seq), "") == bits, "stream %d slot %d flush %d" % (s, slot, seq)
                n_bits += len(bits)
    assert n_bits > 30000


def _all_streams_equal(ev, iq, types, thresh, all_flushes=True, wide=0, orc=None):
    """every stream of the batch against the oracle (OpenMP, one receiver per stream): vectorised comparison.
    orc: the oracle's events if they were computed already (one ORC_EVENT_DTYPE array per stream)"""
    if orc is None:
        orc = O.process_many(iq, types, thresh, wide)
    gs, gm = api.events_canon(ev)
    order = np.argsort(gs, kind="stable")  # (several drains concatenated: each is ordered by stream)
    gs, gm = gs[order], gm[order]
    bounds = np.searchsorted(gs, np.arange(len(orc) + 1))
    minb = np.array([10, 7, 7, 7, 11])
    total = 0
    for s in range(len(orc)):
        e = orc[s]
        if not all_flushes:
            e = e[(e["byte_cnt"] >= minb[e["slot"]]) & ~((e["slot"] == 3) & (e["byte_cnt"] >= 64)) & ~((e["slot"] == 4) & (e["byte_cnt"] > 60))]
        wm = O.canon(e)
        wm = wm[np.lexsort((wm[:, 1], wm[:, 0]))]
        g = gm[bounds[s]:bounds[s + 1]]
        g = g[np.lexsort((g[:, 1], g[:, 0]))]
        assert g.shape == wm.shape and np.array_equal(g, wm), "stream %d" % s
        total += len(wm)
    return total


def test_config1_single_stream_tfa123_48_blocks():
    """BASELINE configs[1], exact shape: ONE 1.536 MS/s stream, TFA_1/2/3 concurrent (-T 7), 48 blocks -- every flush."""
    iq = synth.gen_batch(77, 5, 1, 48)
    with api.Receiver(1, 0x07, 500, 0, max_blocks=48, all_flushes=True) as r:
        r.submit(iq)
        ev = r.drain()
        assert _all_streams_equal(ev, iq, 0x07, 500) > 30
        assert sorted(set(ev["slot"].tolist())) == [0, 1, 2]
        assert r.fm_stats()["host_mismatch"] == 0


def test_quarter_of_config2_every_stream_against_the_oracle():
    """256 streams x 48 blocks x all five protocols (a quarter of BASELINE configs[2]; bench.py's gate checks all 1024 on
    every run): EVERY stream's complete flush log against the oracle, two submits (24 + 24 blocks) in flight."""
    n_streams, n_blocks = 256, 48
    iq = synth.gen_batch(1000, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=24, all_flushes=True, max_events=n_streams * 24 * 40) as r:
        r.submit(np.ascontiguousarray(iq[:, :24 * 65536]))
        r.submit(np.ascontiguousarray(iq[:, 24 * 65536:]))
        ev = np.concatenate([r.drain(), r.drain()])
        total = _all_streams_equal(ev, iq, 0x2F, 500)
        assert total > 80 * n_streams
        assert r.fm_stats()["host_mismatch"] == 0


def test_config2_full_size_every_stream():
    """BASELINE configs[2] at its full size: 1024 streams x 48 blocks x all five protocols in ONE submit (the shape
    bench.py times), every stream's complete flush log (TFREC_AMD_F_ALL_FLUSHES) against the oracle."""
    n_streams, n_blocks = 1024, 48
    iq = synth.gen_batch(1000, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, max_events=n_streams * n_blocks * 40) as r:
        r.submit(iq)
        ev = r.drain()
        total = _all_streams_equal(ev, iq, 0x2F, 500)
        assert total > 80 * n_streams
        assert r.fm_stats()["host_mismatch"] == 0
        assert r.stats()["biquad_segments"] > 40 * n_streams  # (256-slot segments: ~46 per stream and submit)


def test_hostile_and_degenerate_inputs_over_ragged_submits():
    """Inputs a receiver must survive, each against the oracle, in one batch cut into submits of 1, 2, 5 and 3 blocks: pure
    silence (no window at all), full-scale DC (0x00 / 0xff: the trigger never releases, the discriminator sits on its
    axis / diagonal constants), uniform random bytes (every demodulator's window open for the whole stream, garbage
    bits, a window that spans all submits), a periodic pattern (one discriminator direction repeated), a real burst
    stream, and a stream that goes silent half way."""
    n_blocks = 11
    rng = np.random.default_rng(5)
    good = synth.gen_batch(61, 0, 2, n_blocks)
    n = good.shape[1]
    rows = [np.full(n, 0x80, np.uint8), np.zeros(n, np.uint8), np.full(n, 0xFF, np.uint8),
            rng.integers(0, 256, n, dtype=np.uint8), np.tile(np.array([0x90, 0x70, 0x60, 0xA0, 0x85, 0x7B], np.uint8), n // 6 + 1)[:n],
            good[0], good[1].copy()]
    rows[6][n // 2:] = 0x80
    iq = np.stack(rows)
    cuts = [0, 1, 3, 8, 11]
    with api.Receiver(len(iq), 0x2F, 500, 0, max_blocks=5, all_flushes=True, max_events=1 << 16) as r:
        evs = []
        for a, b in zip(cuts, cuts[1:]):
            r.submit(np.ascontiguousarray(iq[:, a * 65536:b * 65536]))
            evs.append(r.drain())
        ev = np.concatenate(evs)
        for s in range(len(iq)):
            o = oracle_events(iq[s], 0x2F, 500)
            check_stream(ev, s, o)
        assert len(api.event_tuples(ev, 0)) == 0  # silence: nothing at all
        assert r.fm_stats()["host_mismatch"] == 0


def test_event_buffer_overflow_is_reported_and_the_context_goes_on():
    """max_events too small: drain returns the events that fit with TFREC_AMD_E_OVERFLOW (include/tfrec_amd.h); state and
    flush ordinals move on, so the next submit's events are the oracle's again."""
    n_streams, n_blocks = 4, 16
    iq = synth.gen_batch(62, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=8, all_flushes=True, max_events=16) as r:
        r.submit(np.ascontiguousarray(iq[:, :8 * 65536]))
        with pytest.raises(api.TfrecAmdError) as ei:
            r.drain()
        assert ei.value.code == api.E_OVERFLOW
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=8, all_flushes=True, max_events=16) as r:
        r.submit(np.ascontiguousarray(iq[:, :8 * 65536]))
        first = r.drain(allow_overflow=True)
        assert len(first) == 16
    # (a context with room: the reference for the second half's events)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=8, all_flushes=True) as r:
        r.submit(np.ascontiguousarray(iq[:, :8 * 65536]))
        r.submit(np.ascontiguousarray(iq[:, 8 * 65536:]))
        ev = np.concatenate([r.drain(), r.drain()])
        for s in range(n_streams):
            check_stream(ev, s, oracle_events(iq[s], 0x2F, 500))


@pytest.mark.parametrize("deep", ["1", "0"])
@pytest.mark.parametrize("every", [1, 3])
def test_whb_speculation_failures_are_redone_exactly(every, deep, monkeypatch):
    """WHB stage 2 speculates its decision levels and whb_verify_kernel checks them against the exact recurrence; a stream
    that fails is redone by the exact kernel from the state the submit started from, its speculative events retracted, and
    every later submit that was speculated from the superseded state is redone too.  TFREC_AMD_WHB_FORCE_FAIL declares
    every N-th (stream + submit) failed: the events must still be the oracle's -- with four submits in flight (a failed
    submit's successors are already running from the wrong state), alternating submit / drain, and in both stream layouts."""
    monkeypatch.setenv("TFREC_AMD_WHB_FORCE_FAIL", str(every))
    monkeypatch.setenv("TFREC_AMD_DEEP", deep)
    n_streams, n_blocks = 24, 40
    iq = synth.gen_batch(91, 3, n_streams, n_blocks)
    parts = [np.ascontiguousarray(iq[:, a * 65536:b * 65536]) for a, b in ((0, 6), (6, 9), (9, 20), (20, 28), (28, 33), (33, 40))]
    for types in (0x2F, 0x20):
        with api.Receiver(n_streams, types, 500, 0, max_blocks=11, all_flushes=True, experiments=True) as r:
            evs, q = [], 0
            for k in range(len(parts)):
                while q < len(parts) and q - k < api.FIFO_DEPTH:
                    r.submit(parts[q])
                    q += 1
                evs.append(r.drain())
            ev = np.concatenate(evs)
            assert not (ev["status"] == 0xFF).any()
            assert _all_streams_equal(ev, iq, types, 500) > (10 if types == 0x2F else 2) * n_streams
            redone = r.stats()["whb_respeculated"]
            assert redone >= n_streams * len(parts) // every // 2, redone
        with api.Receiver(n_streams, types, 500, 0, max_blocks=11, all_flushes=False, experiments=True) as r:  # default mode, one in flight
            evs = []
            for p in parts:
                r.submit(p)
                evs.append(r.drain())
            assert _all_streams_equal(np.concatenate(evs), iq, types, 500, all_flushes=False) > n_streams


def test_whb_speculation_is_verified_and_rarely_fails():
    """Without forced failures: the speculative WHB stage reproduces the exact kernel's events (TFREC_AMD_WHB_EXACT=1 is the
    wave-per-stream recurrence, exact by itself) and the verification accepts practically every stream."""
    n_streams, n_blocks = 64, 24
    iq = synth.gen_batch(92, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x20, 500, 0, max_blocks=n_blocks, all_flushes=True) as r:
        r.submit(iq)
        ev = r.drain()
        assert _all_streams_equal(ev, iq, 0x20, 500) > n_streams
        assert r.stats()["whb_respeculated"] <= 1


def test_fm_dev_nrzs_probe_equals_the_real_reference(golden_dir):
    """fm_dev_nrzs (dsp_stuff.cpp:269-279) on the device, including the +-1e9 clamp (101 of the golden inputs reach it;
    no int16 sample pair does): the probes of the REAL reference function (tests/golden/unit_probes.npz)."""
    z = np.load(os.path.join(golden_dir, "unit_probes.npz"))
    got = api.fm_dev_nrzs_probe(z["fm_in"])
    want = z["fm_out"][:, 1]
    assert np.array_equal(got, want)
    assert (np.abs(want) == 1000000000).sum() >= 100


# ---------------------------------------------------------------------------------------------------------------------
# The state bench.py TIMES: 1024 streams x 48 blocks per submit, the FIFO kept four deep, decoder / biquad / slicer / FIR
# state carried over five DIFFERENT batches (decoder.cpp:118-122 start() rebase, tfa2.cpp:325-334 "last_bit_idx and the
# biquad state are not reset", whb.cpp:616-623) -- every stream's concatenated flush log against the oracle run over the
# same 240 blocks as ONE long stream per receiver.
_STEADY = {}


def _steady_inputs():
    """1024 streams of 240 blocks (5.1 s of signal each, bursts all the way through: windows span the batch boundaries),
    cut into five 48-block batches; the oracle over each stream's 240 blocks in one go."""
    if not _STEADY:
        n_streams, n_blocks, n_batches = 1024, 48, 5
        long_iq = synth.gen_batch(4000, 0, n_streams, n_blocks * n_batches)
        row = n_blocks * 65536
        _STEADY["batches"] = [long_iq[:, k * row:(k + 1) * row] for k in range(n_batches)]  # views
        _STEADY["orc"] = O.process_many(long_iq, 0x2F, 500, 0, cap=2048)
    return _STEADY["batches"], _STEADY["orc"]


@pytest.mark.parametrize("force_fail", [0, 3], ids=["plain", "whb_every_third_stream_redone"])
def test_config2_steady_state_five_batches_vs_oracle(force_fail, monkeypatch):
    """BASELINE configs[2] in the state the benchmark times it: five different 1024 x 48-block batches through ONE
    context with four submits in flight at all times (submit k+4 is queued before submit k is drained), all flushes, every
    stream of every batch against the oracle run over the 240 blocks as one stream.  Second variant: every third
    (stream + submit) fails its WHB check on purpose, so the exact redo of submit k (milliseconds, 341 streams) runs while
    the speculative kernels of submits k+1 .. k+3 work on the same streams' live state (the redo must not work
    in place)."""
    import torch

    if force_fail:
        monkeypatch.setenv("TFREC_AMD_WHB_FORCE_FAIL", str(force_fail))
    batches, orc = _steady_inputs()
    n_streams, n_blocks = 1024, 48
    dev = [torch.from_numpy(np.ascontiguousarray(b)).to("cuda:0") for b in batches]
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True, max_events=n_streams * n_blocks * 40, experiments=True) as r:
        evs, q = [], 0
        for k in range(len(dev)):
            while q < len(dev) and q - k < api.FIFO_DEPTH:
                r.submit(dev[q])
                q += 1
            evs.append(r.drain())
        ev = np.concatenate(evs)
        assert not (ev["status"] == 0xFF).any()
        total = _all_streams_equal(ev, None, 0x2F, 500, orc=orc)
        assert total > 5 * 80 * n_streams
        assert r.fm_stats()["host_mismatch"] == 0
        redone = r.stats()["whb_respeculated"]
        if force_fail:
            assert redone >= n_streams * len(dev) // force_fail
        else:  # (a handful of real speculation failures: windows locked across a batch boundary with an ambiguous candidate)
            assert redone <= 64, redone
    # the batches really continue windows of the one before: flushes of windows that were open across a batch boundary
    # (they fire less than the shortest window timeout after it) exist behind every boundary
    M = n_blocks * 8192
    for k in range(1, len(dev)):
        early = sum(int(((e["end_sample"] >= k * M) & (e["end_sample"] < k * M + 300)).sum()) for e in orc)
        assert early > 0, k


def test_wide_filter_events_match_oracle():
    """-W (dec_filter_taps1w, dsp_stuff.cpp:91-117, 176-178) down to the flush events, deterministic: 24 streams x 24
    blocks, all five protocols, every flush, two submits."""
    n_streams, n_blocks = 24, 24
    iq = synth.gen_batch(314, 0, n_streams, n_blocks)
    with api.Receiver(n_streams, 0x2F, 500, 1, max_blocks=16, all_flushes=True) as r:
        r.submit(np.ascontiguousarray(iq[:, :16 * 65536]))
        r.submit(np.ascontiguousarray(iq[:, 16 * 65536:]))
        ev = np.concatenate([r.drain(), r.drain()])
        total = _all_streams_equal(ev, iq, 0x2F, 500, wide=1)
        assert total > 40 * n_streams
        assert sorted(set(ev["slot"].tolist())) == [0, 1, 2, 3, 4]
        assert r.fm_stats()["host_mismatch"] == 0
    # (the wide filter is a different receiver: the narrow one's events differ on the same input)
    narrow = O.process_many(iq[:4], 0x2F, 500, 0)
    wide = O.process_many(iq[:4], 0x2F, 500, 1)
    assert any(len(a) != len(b) or not np.array_equal(a, b) for a, b in zip(narrow, wide))


@pytest.mark.parametrize("perturb", [3000, -20000, 150000])
def test_whb_frozen_average_off_by_some_is_carried_into_a_redo(perturb, monkeypatch):
    """A WHB window that is locked and still open at a submit boundary, whose speculated frozen decision level is NOT the
    exact integer: the check accepts it if no candidate test could tell the two apart and carries the difference to the
    next submit; if that submit then fails, the exact redo restores the snapshot (speculated integer) and must continue
    the window with the exact one.  TFREC_AMD_WHB_TEST_PERTURB=D makes the speculative kernel freeze
    (int)avg + D with the ambiguity rule widened to |D|; every second (stream + submit) is failed on purpose, so a passed
    submit with a carry is always followed by a redo.  Submits of 1-3 blocks: most telegrams span a boundary."""
    monkeypatch.setenv("TFREC_AMD_WHB_TEST_PERTURB", str(perturb))
    monkeypatch.setenv("TFREC_AMD_WHB_FORCE_FAIL", "2")
    n_streams, n_blocks = 48, 36
    iq = synth.gen_batch(2718, 0, n_streams, n_blocks)
    cuts = [0]
    rng = np.random.default_rng(11)
    while cuts[-1] < n_blocks:
        cuts.append(min(n_blocks, cuts[-1] + int(rng.integers(1, 4))))
    parts = [np.ascontiguousarray(iq[:, a * 65536:b * 65536]) for a, b in zip(cuts, cuts[1:])]
    with api.Receiver(n_streams, 0x20, 500, 0, max_blocks=3, all_flushes=True, experiments=True) as r:
        evs, q = [], 0
        for k in range(len(parts)):
            while q < len(parts) and q - k < api.FIFO_DEPTH:
                r.submit(parts[q])
                q += 1
            evs.append(r.drain())
        ev = np.concatenate(evs)
        assert not (ev["status"] == 0xFF).any()
        assert _all_streams_equal(ev, iq, 0x20, 500) > 2 * n_streams
        assert r.stats()["whb_respeculated"] >= n_streams * len(parts) // 4
