"""Host-side C++ mirror of the reference's plugin surface (tfrec_amd/host): byte-level telegrams through
decoder::store_bytes + decoder::flush -- the reference's own '-X' test entry (main.cpp:24-53) -- against the
golden outputs of the REAL reference decoders; and, on a GPU box, the batched '-L' replay CLI end to end."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tfrec_amd", "host")
CLI = os.path.join(HOST, "tfrec_gpu")


@pytest.fixture(scope="module")
def cli():
    from tfrec_amd import _build
    _build.build_device_lib()
    subprocess.check_call(["make", "-s", "-C", HOST])
    return CLI


def test_byte_level_telegrams_like_reference_dash_X(cli, golden_dir, tmp_path):
    cases = json.load(open(os.path.join(golden_dir, "kat_bytes.json")))["cases"]
    for c in cases:
        f = tmp_path / "kat.txt"
        f.write_text(c["hex"] + "\n")
        out = subprocess.run([cli, "-T", "%x" % c["types"], "-X", str(f)], capture_output=True, text=True, check=True,
                             env=dict(os.environ, TFREC_HOST_RECORDS="1")).stdout
        lines = [ln for ln in out.splitlines() if ln.strip()]
        text = [ln for ln in lines if not ln.startswith("D ")]
        recs = [ln.split() for ln in lines if ln.startswith("D ")]
        assert text == c["text"], c["hex"]
        want = c["data"]
        assert len(recs) == len(want), c["hex"]
        for r, w in zip(recs, want):
            # golden: [slot, type, id_hex, temp_hex, hum_hex, seq, alarm, rssi, flags]
            assert int(r[1]) == w[1] and int(r[2], 16) == int(w[2], 16)
            assert float.fromhex(r[3]) == float.fromhex(w[3]) and float.fromhex(r[4]) == float.fromhex(w[4])
            assert [int(x) for x in r[5:9]] == w[5:9]


def test_readme_vector(cli, tmp_path):
    f = tmp_path / "readme.txt"
    f.write_text("# README.md:123\n2d d4 65 b0 86 20 23 60 e0 56 97\n")
    out = subprocess.run([cli, "-T", "1", "-X", str(f)], capture_output=True, text=True, check=True).stdout
    assert "TFA1 ID 65b0 +22.0 35% seq e lowbat 0 RSSI 0" in out


@pytest.mark.gpu
def test_dump_replay_cli_matches_reference_text(cli, golden_dir, tmp_path):
    from tfrec_amd import synth
    cases = json.load(open(os.path.join(golden_dir, "streams.json")))["cases"]
    c = cases[0]  # 48 blocks, all five protocols, -t 500: stdout of the real reference is the golden text
    iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
    p = tmp_path / "s.iq"
    iq.tofile(p)
    # a second, shorter stream in the same batch must not disturb the first
    c2 = cases[2]
    iq2 = synth.gen_stream(c2["seed"], c2["stream"], c2["n_blocks"], c2["proto_mask"], c2["noise_q8"])
    p2 = tmp_path / "s2.iq"
    iq2.tofile(p2)
    out = subprocess.run([cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", "7", "-L", str(p)],
                         capture_output=True, text=True, check=True).stdout
    want = [ln for ln in c["text"].splitlines() if ln != "Inverted SYNC"]
    assert [ln for ln in out.splitlines() if ln.strip()] == want
    out2 = subprocess.run([cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-L", str(p), "-L", str(p2)],
                          capture_output=True, text=True, check=True).stdout
    assert len([ln for ln in out2.splitlines() if ln.strip()]) >= len(want)


def test_batched_sink_and_per_record_handler(cli, tmp_path):
    """-E starts the handler once and feeds it '<stream> <reference handler args>' lines on stdin;
    -e keeps the reference's one-exec-per-record contract (decoder.cpp:67-96)."""
    f = tmp_path / "kat.txt"
    f.write_text("# README.md:123\n2d d4 65 b0 86 20 23 60 e0 56 97\n2d d4 65 b0 86 20 23 60 e0 56 97\n")
    sink = tmp_path / "sink.out"
    subprocess.run([cli, "-T", "1", "-q", "-X", str(f), "-E", "cat > %s" % sink], check=True)
    lines = sink.read_text().splitlines()
    assert len(lines) == 2  # (the -X entry uses a fresh flush per line: both records are delivered)
    for ln in lines:
        p = ln.split()
        assert p[:8] == ["0", "65b0", "+22.0", "35", "14", "0", "0", "0"] and int(p[8]) > 0
    out = subprocess.run([cli, "-T", "1", "-q", "-X", str(f), "-e", "echo REC"], capture_output=True, text=True,
                         check=True).stdout
    recs = [ln.split() for ln in out.splitlines() if ln.startswith("REC")]
    assert len(recs) == 2 and recs[0][1:8] == ["65b0", "+22.0", "35", "14", "0", "0", "0"]


@pytest.mark.gpu
def test_batched_sink_on_dump_replay(cli, golden_dir, tmp_path):
    from tfrec_amd import synth
    cases = json.load(open(os.path.join(golden_dir, "streams.json")))["cases"]
    c = cases[0]
    files = []
    for k in range(3):  # three streams in one batch: records carry their stream index
        iq = synth.gen_stream(c["seed"], c["stream"] + k, c["n_blocks"], c["proto_mask"], c["noise_q8"])
        p = tmp_path / ("s%d.iq" % k)
        iq.tofile(p)
        files += ["-L", str(p)]
    sink = tmp_path / "sink.out"
    out = subprocess.run([cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", "16", "-E", "cat > %s" % sink] + files,
                         capture_output=True, text=True, check=True).stdout
    recs = [ln.split() for ln in sink.read_text().splitlines()]
    assert len(recs) >= 20 and {r[0] for r in recs} == {"0", "1", "2"}
    # every record corresponds to a printed telegram (TX22 / WHB telegrams expand to several records)
    assert len(recs) >= len([ln for ln in out.splitlines() if ln.strip()])


def _three_streams(golden_dir, tmp_path):
    from tfrec_amd import synth
    g = json.load(open(os.path.join(golden_dir, "handler_records.json")))
    c = g["case"]
    files = []
    for k in range(3):
        iq = synth.gen_stream(c["seed"], c["stream"] + k, c["n_blocks"], c["proto_mask"], c["noise_q8"])
        p = tmp_path / ("s%d.iq" % k)
        iq.tofile(p)
        files.append(str(p))
    return g, c, files


@pytest.mark.gpu
def test_handler_records_equal_the_real_reference(cli, golden_dir, tmp_path):
    """what the result sink delivers == the command lines the REAL reference's execute_handler builds
    (decoder.cpp:67-96) after store_data's dedupe (:46-65, incl. the WHB sequence rule), for three streams in one batch;
    -m 1: the summary of flush_storage (:98-109) in the reference's order.  Goldens: oracle/mint_handler_records.py."""
    g, c, files = _three_streams(golden_dir, tmp_path)
    largs = sum((["-L", f] for f in files), [])
    base = [cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-q", "-b", "16"]
    for mode in (0, 1):
        sink = tmp_path / ("sink%d.out" % mode)
        subprocess.run(base + ["-m", str(mode), "-E", "cat > %s" % sink] + largs, check=True)
        recs = [ln.split() for ln in sink.read_text().splitlines()]
        for k in range(3):
            got = [" ".join(r[1:-1]) for r in recs if r[0] == str(k)]  # minus the stream tag and ts
            assert got == g["streams"][k]["mode%d" % mode], (mode, k)
    # the reference's own one-exec-per-record path on one stream
    out = subprocess.run(base + ["-e", "echo REC", "-L", files[1]], capture_output=True, text=True, check=True).stdout
    assert [" ".join(ln.split()[1:-1]) for ln in out.splitlines() if ln.startswith("REC ")] == g["streams"][1]["mode0"]


@pytest.mark.gpu
def test_streams_sharded_over_several_device_contexts(cli, golden_dir, tmp_path):
    """SURVEY 8e on the C++ side: '-d a,b,..' = one host thread + context per entry, dump files sharded by index, events
    concatenated on the host: stdout and the sink are those of a single-context run (two contexts on device 0 here)."""
    g, c, files = _three_streams(golden_dir, tmp_path)
    largs = sum((["-L", f] for f in files), [])
    base = [cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", "7"]
    outs = []
    for dev in ("0", "0,0", "0,0,0"):
        sink = tmp_path / ("sink_%s.out" % dev.replace(",", "_"))
        o = subprocess.run(base + ["-d", dev, "-E", "cat > %s" % sink] + largs, capture_output=True, text=True, check=True).stdout
        outs.append((o, [ln.split()[:-1] for ln in sink.read_text().splitlines()]))
    assert len(outs[0][0].splitlines()) > 50
    assert outs[1] == outs[0] and outs[2] == outs[0]


@pytest.mark.gpu
def test_reference_linked_cli_replays_a_dump(golden_dir, tmp_path):
    """The adapter linked with the REAL reference decoders (oracle/_ref/tfrec_gpu_ref, built where the reference tree
    exists; it travels with the snapshot): GPU flush events -> unchanged tfa1/tfa2/whb_decoder::flush -> the reference's
    own printf -- the text the reference prints for the same dump."""
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "tfrec_gpu_ref")
    if not os.path.exists(ref_cli):
        pytest.skip("oracle/_ref/tfrec_gpu_ref was not built (no reference tree at build time)")
    from tfrec_amd import synth
    c = json.load(open(os.path.join(golden_dir, "streams.json")))["cases"][0]
    iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
    p = tmp_path / "s.iq"
    iq.tofile(p)
    out = subprocess.run([ref_cli, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", "7", "-L", str(p)],
                         capture_output=True, text=True, check=True).stdout
    want = [ln for ln in c["text"].splitlines() if ln != "Inverted SYNC"]
    got = [ln for ln in out.splitlines() if ln.strip() and not ln.startswith("WHB: Samples")]
    assert got == want


@pytest.mark.gpu
def test_bits_replay_reproduces_what_store_bit_prints(cli, golden_dir, tmp_path):
    """SURVEY 8b, BITS mode: `tfrec_gpu -B` hands every demodulated bit to the decoder's own store_bit and replays every
    flush; stdout is then the reference's complete text INCLUDING the "Inverted SYNC" lines of tfa2_decoder::store_bit
    (tfa2.cpp:294-300) -- for the mirror decoders and for the adapter linked with the reference's unchanged objects.
    Streams with I/Q swapped (inverted FSK polarity), goldens from the real reference (oracle/mint_inverted.py)."""
    from tfrec_amd import synth
    cases = json.load(open(os.path.join(golden_dir, "inverted_sync.json")))["cases"]
    clis = [cli]
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "tfrec_gpu_ref")
    if os.path.exists(ref_cli):
        clis.append(ref_cli)
    files = []
    for k, c in enumerate(cases):
        iq = synth.gen_stream(c["seed"], c["stream"], c["n_blocks"], c["proto_mask"], c["noise_q8"])
        iq = iq.reshape(-1, 2)[:, ::-1].reshape(-1).copy()
        p = tmp_path / ("inv%d.iq" % k)
        iq.tofile(p)
        files.append(str(p))
    for exe in clis:
        for c, f in zip(cases, files):
            want = [ln for ln in c["text"].splitlines() if ln.strip()]
            assert want.count("Inverted SYNC") >= 3
            for blocks in ("5", "24"):  # windows cut by submits / one submit
                out = subprocess.run([exe, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", blocks, "-B", "-L", f],
                                     capture_output=True, text=True, check=True).stdout
                got = [ln for ln in out.splitlines() if ln.strip() and not ln.startswith("WHB: Samples")]
                assert got == want, (exe, blocks)
            # the byte-level replay prints the same telegrams, without store_bit's own line
            out = subprocess.run([exe, "-T", "%x" % c["types"], "-t", str(c["thresh"]), "-b", "5", "-L", f],
                                 capture_output=True, text=True, check=True).stdout
            got = [ln for ln in out.splitlines() if ln.strip() and not ln.startswith("WHB: Samples")]
            assert got == [ln for ln in want if ln != "Inverted SYNC"]


def test_debug_diagnostics_equal_the_real_reference(cli, golden_dir, tmp_path):
    """-q / -D / -D -D: the candidate lines ('#NNN <time> <bytes>'), BAD lines and history entries of tfa1.cpp:50-55,
    tfa2.cpp:77-82,133-141,152-157,195-202 and whb.cpp:306-311,344-348,381-384,413-418,488-493,551-558, against what the
    real reference printed for the same bytes (tests/golden/kat_debug.json, wall-clock second masked)."""
    import re
    cases = json.load(open(os.path.join(golden_dir, "kat_debug.json")))["cases"]
    flags = {"-1": ["-q"], "1": ["-D"], "2": ["-D", "-D"]}
    for c in cases:
        f = tmp_path / "katd.txt"
        f.write_text(c["hex"] + "\n")
        for lvl, fl in flags.items():
            out = subprocess.run([cli, "-T", "%x" % c["types"]] + fl + ["-X", str(f)], capture_output=True, text=True,
                                 check=True).stdout
            assert re.sub(r"^(#\d{3}) \d+ ", r"\1 T ", out, flags=re.M) == c["text"][lvl], (c["hex"], lvl)
