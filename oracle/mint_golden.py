#!/usr/bin/env python3
"""Mint golden vectors from the REAL reference and pin the C restatement against it.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (this container): builds
oracle/_ref/ref_driver from the reference's own sources (oracle/Makefile), drives it on synthetic IQ
streams and byte-level telegrams, compares every observable with oracle/tfrec_oracle.c, and writes the
fixtures the CPU tests replay on any machine:

  tests/golden/kat_bytes.json     byte-level known answers (README.md:123 + SURVEY App. D) through -X
  tests/golden/streams.json       generator parameters + reference events/data/text/hashes
  tests/golden/config5.json       the same for BASELINE config 5 (15.36 MS/s input, 10:1 stage, int16 entry)
  tests/golden/iq_*.npz           small raw IQ captures with the reference's outputs (generator-independent)
  tests/golden/unit_probes.npz    fm_dev / fm_dev_nrzs / iir2::step probes of the reference functions

Usage:  python oracle/mint_golden.py [--campaign N]   (N extra random streams compared, not stored)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tfrec_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# (types mask for -T, telegram bytes)  -- README.md:123 first, then SURVEY.md Appendix D
KAT_BYTES = [
    (0x01, "2d d4 65 b0 86 20 23 60 e0 56 97"),
    (0x01, "2d d4 12 34 81 05 6a e0 30 56 59"),
    (0x01, "2d d4 7f ff 8a aa 7f 60 50 56 f1"),
    (0x01, "2d d4 00 01 83 99 55 50 10 56 73"),
    (0x02, "2d d4 99 06 25 2f e8"),
    (0x02, "2d d4 9a c3 91 7d 32"),
    (0x02, "2d d4 91 40 00 6a c0"),
    (0x04, "2d d4 9f c4 45 30 72"),
    (0x08, "2d d4 a1 52 06 22 10 34 39"),
    (0x08, "2d d4 a3 dd 05 87 10 99 21 23 35 2d 40 7b 1c"),
    (0x20, "4b 2d d4 2b 11 02 11 22 33 44 55 00 10 00 d5 07 f6 c4 3b 95 6c"),
    (0x20, "4b 2d d4 2b 16 03 11 22 33 44 55 40 0a 00 d5 00 33 00 d4 00 34 00 8f e6 dd e2"),
    (0x20, "4b 2d d4 2b 17 04 11 22 33 44 55 00 0a 00 d5 00 33 00 00 d4 00 34 01 1e f7 c7 30"),
    (0x20, "4b 2d d4 2b 19 06 11 22 33 44 55 00 0b 00 d5 07 f6 00 2d 00 d4 07 f5 00 2e 6b 6d b0 50"),
    (0x20, "4b 2d d4 2b 19 09 11 22 33 44 55 00 0b 00 d5 0f 06 00 2d 00 d4 04 4c 00 2e 7a 64 f5 ab"),
    (0x20, "4b 2d d4 2b 1d 07 11 22 33 44 55 00 0c 00 d5 00 2d 07 ce 00 50 00 d4 00 2e 07 cf 00 51 54 88 8a 2b"),
    (0x20, "4b 2d d4 2b 25 08 11 22 33 44 55 00 0d 40 d2 00 09 " + "c3 06 " * 10 + "51 93 60 1e"),
    (0x20, "4b 2d d4 2b 26 0b 11 22 33 44 55 00 03 b6 " + "c2 07 3a 05 " * 6 + "96 d4 61 3c"),
    (0x20, "4b 2d d4 2b 15 10 11 22 33 44 55 00 0e 80 05 20 10 c0 02 00 03 8e 99 ba ad"),
    (0x20, "4b 2d d4 2b 2d 11 11 22 33 44 55 00 0f " + "00 74 00 56 00 de 00 33 00 bc 00 3e 00 dc 00 37 " * 2
     + "ba 1c 40 3d"),
    (0x20, "4b 2d d4 2b 14 12 11 22 33 44 55 00 10 33 35 33 00 00 d7 34 72 bf fd b2"),
    # every decoder sees every telegram (like a live receiver with -T 2f): cross-protocol rejects
    (0x2F, "2d d4 65 b0 86 20 23 60 e0 56 97"),
    (0x2F, "2d d4 a1 52 06 22 10 34 39"),
    (0x2F, "4b 2d d4 2b 16 03 11 22 33 44 55 40 0a 00 d5 00 33 00 d4 00 34 00 8f e6 dd e2"),
    # corrupted CRCs / unknown WHB type / short frames
    (0x01, "2d d4 65 b0 86 20 23 60 e0 56 96"),
    (0x02, "2d d4 99 06 25 2f e9"),
    (0x08, "2d d4 a1 52 06 22 10 34 38"),
    (0x20, "4b 2d d4 2b 11 05 11 22 33 44 55 00 10 00 d5 07 f6 c4 3b 95 6c"),
    (0x20, "4b 2d d4 2b 11 02 11 22 33 44 55 00 10 00 d5 07 f6 c4 3b 95 6d"),
    (0x2F, "2d d4 65"),
]


def data_to_json(recs):
    return [[r[0], r[1], "%x" % r[2], float(r[3]).hex(), float(r[4]).hex(), r[5], r[6], r[7], r[8]] for r in recs]


def events_to_json(evs):
    return [[e[0], e[1], e[2], e[3], e[4], e[5].hex()] for e in evs]


def check(cond, msg):
    if not cond:
        print("MISMATCH:", msg)
        sys.exit(1)


def ref_hex(types, hexline, tmp):
    hp = os.path.join(tmp, "kat.txt")
    ep = os.path.join(tmp, "kat.ev")
    with open(hp, "w") as f:
        f.write(hexline.strip() + "\n")
    out = subprocess.run([O.REF_DRIVER, "hex", "%x" % types, hp, ep], capture_output=True, text=True, check=True)
    text = out.stdout.split("---\n", 1)[1]
    ev, data, _ = O.parse_ref_events(ep)
    lines = [ln for ln in text.splitlines() if ln.strip()]
    return lines, ev, data


def mint_kats(tmp):
    out = []
    for types, hexline in KAT_BYTES:
        lines, ev, data = ref_hex(types, hexline, tmp)
        o = O.Oracle(types)
        o.hex(bytes(int(x, 16) for x in hexline.split()))
        olines = [ln for ln in o.text().splitlines() if ln.strip()]
        check(olines == lines, "KAT text %r: %r vs %r" % (hexline, olines, lines))
        check(o.data() == data, "KAT data %r" % hexline)
        check(o.events() == ev, "KAT events %r" % hexline)
        out.append(dict(types=types, hex=" ".join(hexline.split()), text=lines, data=data_to_json(data),
                        events=events_to_json(ev)))
    check(out[0]["text"] == ["TFA1 ID 65b0 +22.0 35% seq e lowbat 0 RSSI 0"], "README.md:123 vector")
    with open(os.path.join(GOLD, "kat_bytes.json"), "w") as f:
        json.dump(dict(source="oracle/_ref/ref_driver hex (real reference decoders), README.md:123 + SURVEY App. D",
                       cases=out), f, indent=0)
    print("kat_bytes: %d cases pinned" % len(out))


def mint_kat_debug(tmp):
    """The decoders' diagnostics at debug levels -1 (-q), 1 (-D) and 2 (-D -D): the text the REAL reference prints for every
    byte-level known answer (the '#NNN <time> ...' candidate lines, BAD lines, history entries), the wall-clock second
    masked.  The host mirror (tfrec_amd/host/telegram.cpp) is compared with it in tests/test_host_cpp.py."""
    import re
    out = []
    for types, hexline in KAT_BYTES:
        hp = os.path.join(tmp, "katd.txt")
        with open(hp, "w") as f:
            f.write(hexline.strip() + "\n")
        levels = {}
        for lvl in (-1, 1, 2):
            r = subprocess.run([O.REF_DRIVER, "hex", "%x" % types, hp, "", str(lvl)], capture_output=True, text=True, check=True)
            text = r.stdout.split("---\n", 1)[1]
            levels[str(lvl)] = re.sub(r"^(#\d{3}) \d+ ", r"\1 T ", text, flags=re.M)
        out.append(dict(types=types, hex=" ".join(hexline.split()), text=levels))
    with open(os.path.join(GOLD, "kat_debug.json"), "w") as f:
        json.dump(dict(source="oracle/_ref/ref_driver hex <types> <file> '' <level> (real reference decoders); '#NNN <time>' masked to '#NNN T'",
                       cases=out), f, indent=0)
    print("kat_debug: %d cases x 3 debug levels pinned" % len(out))


def compare_stream(iq, types, thresh, wide, tmp, tag, bits=True):
    """Run reference + oracle on one IQ array; returns the reference result after asserting equality."""
    p = os.path.join(tmp, "s.iq")
    iq.tofile(p)
    ref = O.run_reference(p, types, thresh, wide, tmp, bits=bits)
    o = O.Oracle(types, thresh, wide, log_bits=bits, keep_dec=True)
    o.process(iq)
    check(np.array_equal(o.dec(), ref["dec"]), tag + ": decimated samples")
    check(o.events() == ref["events"], tag + ": flush events (%d vs %d)" % (len(o.events()), len(ref["events"])))
    check(o.data() == ref["data"], tag + ": store_data records")
    check(o.text() == ref["text"], tag + ": telegram text")
    if bits:
        check(o.bits_text() == ref["bits"], tag + ": store_bit log")
    return ref


STREAM_CASES = [
    # seed, stream, blocks, proto_mask, noise_q8, types, thresh, wide
    dict(seed=7, stream=0, n_blocks=48, proto_mask=0x1F, noise_q8=256, types=0x2F, thresh=500, wide=0),
    dict(seed=7, stream=1, n_blocks=48, proto_mask=0x1F, noise_q8=256, types=0x07, thresh=500, wide=0),
    dict(seed=7, stream=2, n_blocks=24, proto_mask=0x01, noise_q8=256, types=0x01, thresh=500, wide=0),
    dict(seed=7, stream=3, n_blocks=48, proto_mask=0x1F, noise_q8=512, types=0x2F, thresh=0, wide=0),
    dict(seed=7, stream=4, n_blocks=24, proto_mask=0x1F, noise_q8=256, types=0x2F, thresh=500, wide=1),
    dict(seed=11, stream=5, n_blocks=24, proto_mask=0x10, noise_q8=256, types=0x20, thresh=500, wide=0),
    dict(seed=11, stream=6, n_blocks=16, proto_mask=0x0E, noise_q8=768, types=0x2F, thresh=300, wide=0),
    dict(seed=11, stream=7, n_blocks=8, proto_mask=0x00, noise_q8=256, types=0x2F, thresh=500, wide=0),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mint_streams(tmp):
    cases = []
    for c in STREAM_CASES:
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
        ref = compare_stream(iq, c["types"], c["thresh"], c["wide"], tmp, "stream case %r" % c)
        d = dict(c)
        d.update(iq_sha256=sha(iq), dec_sha256=sha(ref["dec"]), events=events_to_json(ref["events"]),
                 data=data_to_json(ref["data"]), text=ref["text"])
        cases.append(d)
        print("stream case seed=%d stream=%d: %d flushes, %d records, %d text lines" % (
            c["seed"], c["stream"], len(ref["events"]), len(ref["data"]), len(ref["text"].splitlines())))
    with open(os.path.join(GOLD, "streams.json"), "w") as f:
        json.dump(dict(source="oracle/_ref/ref_driver run (real reference hot path) on tfrec_amd.synth streams",
                       cases=cases), f, indent=0)


CONFIG5_CASES = [
    # 15.36 MS/s streams (generator rate_mult 10) -> oracle.decim10 -> the real reference's int16 entry (run16)
    dict(seed=31, stream=0, n_blocks=12, proto_mask=0x1F, noise_q8=256, types=0x2F, thresh=500, wide=0),
    dict(seed=31, stream=1, n_blocks=12, proto_mask=0x1F, noise_q8=512, types=0x2F, thresh=500, wide=1),
]


def mint_config5(tmp):
    """BASELINE config 5.  The 10:1 stage has no reference counterpart (its output is pinned only as a hash of
    the C restatement's own result); everything after it is the real reference fed with that int16 stream."""
    cases = []
    for c in CONFIG5_CASES:
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"], rate_mult=10)
        x16 = O.decim10(iq)
        p = os.path.join(tmp, "c5.s16")
        x16.tofile(p)
        ref = O.run_reference(p, c["types"], c["thresh"], c["wide"], tmp, bits=True, in16=True)
        o = O.Oracle(c["types"], c["thresh"], c["wide"], log_bits=True, keep_dec=True)
        o.process_s16(x16)
        tag = "config5 case %r" % c
        check(np.array_equal(o.dec(), ref["dec"]), tag + ": decimated samples")
        check(o.events() == ref["events"], tag + ": flush events")
        check(o.data() == ref["data"], tag + ": store_data records")
        check(o.text() == ref["text"], tag + ": telegram text")
        check(o.bits_text() == ref["bits"], tag + ": store_bit log")
        d = dict(c)
        d.update(iq_sha256=sha(iq), stage0_sha256=sha(x16), dec_sha256=sha(ref["dec"]), events=events_to_json(ref["events"]),
                 data=data_to_json(ref["data"]), text=ref["text"])
        cases.append(d)
        print("config5 case seed=%d stream=%d: %d flushes, %d text lines" % (c["seed"], c["stream"], len(ref["events"]),
                                                                             len(ref["text"].splitlines())))
    with open(os.path.join(GOLD, "config5.json"), "w") as f:
        json.dump(dict(source="tfrec_amd.synth rate_mult=10 -> oracle.decim10 (defined here) -> oracle/_ref/ref_driver "
                              "run16 (real reference hot path on int16 input)", cases=cases), f, indent=0)


def mint_iq_fixtures(tmp):
    """Short raw captures around one burst per protocol, stored with the reference's outputs."""
    for proto in range(5):
        iq = synth.gen_stream(23 + proto, proto, 8, 1 << proto, 256)
        nb = 4 if proto != 4 else 6
        iq = iq[: nb * 65536].copy()  # burst starts at sample 40000 -> inside the kept blocks
        ref = compare_stream(iq, 0x2F, 500, 0, tmp, "iq fixture proto %d" % proto)
        np.savez_compressed(os.path.join(GOLD, "iq_%s.npz" % synth.PROTO_NAMES[proto].lower()), iq=iq,
                            dec=ref["dec"],
                            meta=json.dumps(dict(types=0x2F, thresh=500, wide=0, events=events_to_json(ref["events"]),
                                                 data=data_to_json(ref["data"]), text=ref["text"])))
        print("iq fixture %s: %d bytes IQ, %d flushes, text %r" % (synth.PROTO_NAMES[proto], iq.size,
                                                                    len(ref["events"]), ref["text"].strip()[:60]))


def mint_unit_probes():
    rng = np.random.default_rng(12345)
    # discriminators: random int16-range quads + exact octant/axis cases
    q = rng.integers(-9000, 9001, size=(20000, 4), dtype=np.int32)
    special = []
    for a in (1, 7, 4096, 8191, -3, -8191):
        for (br, bj, ar, aj) in ((a, 0, a, 0), (a, 0, a, a), (a, 0, 0, a), (a, 0, -a, a), (a, 0, -a, 0),
                                 (a, 0, -a, -a), (a, 0, 0, -a), (a, 0, a, -a), (0, 0, 0, 0), (a, a, 0, 0),
                                 (0, 0, a, a)):
            special.append((ar, aj, br, bj))
    q = np.concatenate([np.array(special, dtype=np.int32), q,
                        rng.integers(-32768, 32768, size=(2000, 4), dtype=np.int32)])
    out = subprocess.run([O.REF_DRIVER, "fmdev"], input=q.tobytes(), capture_output=True, check=True).stdout
    r = np.frombuffer(out, dtype=np.int32).reshape(-1, 2)
    L = O.lib()
    mine = np.array([(L.orc_fm_dev(*map(int, x)), L.orc_fm_dev_nrzs(*map(int, x))) for x in q], dtype=np.int32)
    check(np.array_equal(mine, r), "fm_dev / fm_dev_nrzs probes")
    # biquads at the five cut-offs main.cpp:186-217 + tfa2.cpp:321 + whb.cpp:610-611 produce
    cutoffs = [0.5 / (384000 / 17240), 0.5 / (384000 / 9600), 0.5 / (384000 / 8842), 2.0 / 64.0, 0.0025 / 64.0]
    x = np.concatenate([rng.integers(-16384, 16385, size=3000).astype(np.float64),
                        rng.integers(-10 ** 8, 10 ** 8, size=1000).astype(np.float64)])
    iir_out = []
    coeffs = []
    for c in cutoffs:
        out = subprocess.run([O.REF_DRIVER, "iir", repr(c)], input=x.tobytes(), capture_output=True, check=True).stdout
        y = np.frombuffer(out, dtype=np.float64)
        ym = np.empty_like(x)
        L.orc_iir_run(c, x.ctypes.data, ym.ctypes.data, x.size)
        check(np.array_equal(y.view(np.uint64), ym.view(np.uint64)), "iir2 bit-exact at cutoff %r" % c)
        import ctypes as C
        cc = (C.c_double * 5)()
        L.orc_iir_coeffs(c, cc)
        coeffs.append(list(cc))
        iir_out.append(y)
    np.savez_compressed(os.path.join(GOLD, "unit_probes.npz"), fm_in=q, fm_out=r, iir_cutoffs=np.array(cutoffs),
                        iir_in=x, iir_out=np.array(iir_out), iir_coeffs=np.array(coeffs))
    print("unit probes: %d discriminator quads, %d biquad runs bit-exact" % (len(q), len(cutoffs)))
    for c, k in zip(cutoffs, coeffs):
        print("  cutoff %.17g -> %s" % (c, " ".join(float(v).hex() for v in k)))


def campaign(n, tmp):
    rng = np.random.default_rng(99)
    for k in range(n):
        seed = int(rng.integers(1, 1 << 30))
        c = dict(seed=seed, stream=int(rng.integers(0, 1000)), n_blocks=int(rng.choice([8, 16, 32, 48])),
                 proto_mask=int(rng.choice([0x1F, 0x1F, 0x1F, 0x0E, 0x11, 0x01, 0x10])),
                 noise_q8=int(rng.choice([128, 256, 256, 512, 1024, 2048])),
                 types=int(rng.choice([0x2F, 0x2F, 0x07, 0x01, 0x20, 0x0E, 0x28])),
                 thresh=int(rng.choice([500, 500, 0, 200, 100, 1500])), wide=int(rng.choice([0, 0, 0, 1])))
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
        if k % 5 == 4:  # hostile input: uniform random bytes (windows always open, garbage bits)
            iq = rng.integers(0, 256, size=iq.size, dtype=np.uint8)
        ref = compare_stream(iq, c["types"], c["thresh"], c["wide"], tmp, "campaign %r" % c, bits=(k % 3 == 0))
        print("campaign %3d ok: %s -> %d flushes %d records" % (k, c, len(ref["events"]), len(ref["data"])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--campaign", type=int, default=0)
    ap.add_argument("--no-mint", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(O.REFERENCE_DIR):
        print("no /root/reference here: nothing to mint")
        return 1
    O.build()
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        if not a.no_mint:
            mint_kats(tmp)
            mint_kat_debug(tmp)
            mint_unit_probes()
            mint_streams(tmp)
            mint_config5(tmp)
            mint_iq_fixtures(tmp)
        if a.campaign:
            campaign(a.campaign, tmp)
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
