"""CPU tests: the oracle's C restatement against the golden vectors minted from the REAL reference
(oracle/mint_golden.py, run where /root/reference exists).  Bit-exact, no tolerances."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tfrec_amd import synth


def _data(recs):
    return [(r[0], r[1], int(r[2], 16), float.fromhex(r[3]), float.fromhex(r[4]), r[5], r[6], r[7], r[8]) for r in recs]


def _events(evs):
    return [(e[0], e[1], e[2], e[3], e[4], bytes.fromhex(e[5])) for e in evs]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _status_pinned_by_text(o, ref_lines):
    """The per-event verdict the oracle exports (orc_event_t.status) against the REAL reference's output: every flush its
    decoder accepts prints exactly one telegram line (tfa1.cpp:89, tfa2.cpp:169/249, whb.cpp:126-475), per protocol; a
    flush shorter than a telegram is 0, everything else 2."""
    full = o.events_full()
    minb = {0: 10, 1: 7, 2: 7, 3: 7, 4: 11}
    for slot, prefix in ((0, "TFA1 "), (1, "TFA2 "), (2, "TFA3 "), (3, "TX22 "), (4, "WHB")):
        lines = [ln for ln in ref_lines if ln.startswith(prefix) and not ln.startswith("WHB:")]
        assert sum(1 for e in full if e[0] == slot and e[7] == 1) == len(lines), prefix
    for e in full:
        too_short = e[2] < minb[e[0]] or (e[0] == 3 and e[2] >= 64) or (e[0] == 4 and e[2] > 60)
        assert (e[7] == 0) == too_short


def test_readme_known_answer(golden_dir):
    # README.md:123 of the reference: the only in-tree known answer
    o = O.Oracle(0x01)
    o.hex(bytes.fromhex("2dd465b086202360e05697"))
    assert o.text() == "TFA1 ID 65b0 +22.0 35% seq e lowbat 0 RSSI 0\n"


def test_byte_level_known_answers(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "kat_bytes.json")))["cases"]
    assert len(cases) >= 25
    for c in cases:
        o = O.Oracle(c["types"])
        o.hex(bytes(int(x, 16) for x in c["hex"].split()))
        assert [ln for ln in o.text().splitlines() if ln.strip()] == c["text"], c["hex"]
        assert o.data() == _data(c["data"]), c["hex"]
        assert o.events() == _events(c["events"]), c["hex"]
        _status_pinned_by_text(o, c["text"])


def test_unit_probes(golden_dir):
    z = np.load(os.path.join(golden_dir, "unit_probes.npz"))
    L = O.lib()
    q, want = z["fm_in"], z["fm_out"]
    got = np.array([(L.orc_fm_dev(*map(int, x)), L.orc_fm_dev_nrzs(*map(int, x))) for x in q], dtype=np.int32)
    assert np.array_equal(got, want)
    x = np.ascontiguousarray(z["iir_in"])
    for k, c in enumerate(z["iir_cutoffs"]):
        y = np.empty_like(x)
        L.orc_iir_run(float(c), x.ctypes.data, y.ctypes.data, x.size)
        assert np.array_equal(y.view(np.uint64), z["iir_out"][k].view(np.uint64)), "cutoff %r" % c
        cc = (C.c_double * 5)()
        L.orc_iir_coeffs(float(c), cc)
        assert np.array_equal(np.array(list(cc)).view(np.uint64), z["iir_coeffs"][k].view(np.uint64))


def test_synthetic_streams_against_reference_outputs(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "streams.json")))["cases"]
    assert len(cases) >= 6
    for c in cases:
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
        assert _sha(iq) == c["iq_sha256"], "generator drifted: seed %d stream %d" % (c["seed"], c["stream"])
        o = O.Oracle(c["types"], c["thresh"], c["wide"], keep_dec=True)
        assert o.process(iq) == c["n_blocks"]
        assert _sha(o.dec()) == c["dec_sha256"]
        assert o.events() == _events(c["events"])
        assert o.data() == _data(c["data"])
        assert o.text() == c["text"]
        _status_pinned_by_text(o, c["text"].splitlines())


def _inverted_iq(c):
    iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
    return iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy() if c.get("iq_swap") else iq


def test_inverted_polarity_streams_against_reference_outputs(golden_dir):
    """I/Q swapped: the TFA_2-family decoders lock on the complemented sync word (tfa2.cpp:294-300); text incl. the
    "Inverted SYNC" lines, flush events and the store_bit log of the real reference (oracle/mint_inverted.py)."""
    cases = json.load(open(os.path.join(golden_dir, "inverted_sync.json")))["cases"]
    for c in cases:
        iq = _inverted_iq(c)
        assert _sha(iq) == c["iq_sha256"]
        o = O.Oracle(c["types"], c["thresh"], c["wide"], log_bits=True)
        assert o.process(iq) == c["n_blocks"]
        assert o.events() == _events(c["events"])
        assert o.data() == _data(c["data"])
        assert o.text() == c["text"] and c["text"].count("Inverted SYNC") >= 3
        assert o.bits_text() == c["bits"]


def test_config5_int16_entry_against_reference_outputs(golden_dir):
    """BASELINE config 5: 15.36 MS/s input -> 10:1 stage (defined by this project) -> the reference's int16 entry."""
    cases = json.load(open(os.path.join(golden_dir, "config5.json")))["cases"]
    assert len(cases) >= 2
    for c in cases:
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"], rate_mult=10)
        assert _sha(iq) == c["iq_sha256"], "generator drifted: seed %d stream %d" % (c["seed"], c["stream"])
        x16 = O.decim10(iq)
        assert _sha(x16) == c["stage0_sha256"]
        o = O.Oracle(c["types"], c["thresh"], c["wide"], keep_dec=True)
        assert o.process_s16(x16) == c["n_blocks"]
        assert _sha(o.dec()) == c["dec_sha256"]
        assert o.events() == _events(c["events"])
        assert o.data() == _data(c["data"])
        assert o.text() == c["text"]
        assert len(o.events()) >= 10


@pytest.mark.parametrize("name", ["tfa_1", "tfa_2", "tfa_3", "tx22", "whb"])
def test_raw_iq_fixtures(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "iq_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    o = O.Oracle(meta["types"], meta["thresh"], meta["wide"], keep_dec=True)
    o.process(z["iq"])
    assert np.array_equal(o.dec(), z["dec"])
    assert o.events() == _events(meta["events"])
    assert o.data() == _data(meta["data"])
    assert o.text() == meta["text"] and len(meta["text"]) > 0


def test_empty_ragged_and_block_partition():
    o = O.Oracle(0x2F, 500)
    assert o.process(np.zeros(0, dtype=np.uint8)) == 0
    assert o.process(np.full(65535, 128, dtype=np.uint8)) == 0  # partial block dropped (engine.cpp:72-76)
    iq = synth.gen_stream(31, 3, 12)
    whole = O.Oracle(0x2F, 500)
    whole.process(np.concatenate([iq, np.zeros(100, np.uint8)]))  # ragged tail ignored
    parts = O.Oracle(0x2F, 500)
    for a, b in ((0, 1), (1, 6), (6, 12)):
        parts.process(iq[a * 65536:b * 65536])
    assert whole.events() == parts.events() and whole.text() == parts.text() and len(whole.events()) > 0


def test_auto_threshold_moves():
    iq = synth.gen_stream(7, 3, 48, 0x1F, 512)
    o = O.Oracle(0x2F, 0)
    o.process(iq)
    assert o.thresh() != 500  # fm_demod.cpp:58-73 adapted it


@pytest.mark.skipif(not O.have_reference(), reason="real reference only exists in the build container")
def test_live_against_real_reference(tmp_path):
    iq = synth.gen_stream(4242, 17, 16, 0x1F, 384)
    p = str(tmp_path / "s.iq")
    iq.tofile(p)
    ref = O.run_reference(p, 0x2F, 500, 0, str(tmp_path), bits=True)
    o = O.Oracle(0x2F, 500, 0, log_bits=True, keep_dec=True)
    o.process(iq)
    assert np.array_equal(o.dec(), ref["dec"])
    assert o.events() == ref["events"] and o.data() == ref["data"] and o.text() == ref["text"]
    assert o.bits_text() == ref["bits"]
