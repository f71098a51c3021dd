"""'The protocol handlers register unchanged' (north_star), proven by the toolchain: the GPU adapter tfrec_amd/host/
gpu_engine.cpp + main.cpp is compiled against the REFERENCE's own decoder.h / tfa1.h / tfa2.h / whb.h and linked with
the reference's own, unmodified decoder objects + libtfrec_amd.so (oracle/Makefile: _ref/tfrec_gpu_ref).  The byte-level
entry (-X: decoder::store_bytes + flush, main.cpp:24-53 -- the same two calls the adapter uses to hand over a GPU flush
event) then runs HERE, without a GPU, through the real reference decoders.  Only where /root/reference exists."""
import json
import os
import subprocess

import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "tfrec_gpu_ref")
pytestmark = pytest.mark.skipif(not os.path.isdir(O.REFERENCE_DIR), reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def ref_cli():
    from tfrec_amd import _build
    _build.build_device_lib()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    return REF_CLI


def test_adapter_builds_against_the_reference_headers_and_objects(ref_cli):
    assert os.path.exists(ref_cli)
    # the binary really contains the reference's decoders, not the mirror's: the reference's tfa1_decoder has a member the
    # mirror does not (snum, tfa1.h:21) and its translation units carry the reference's file-scope tables
    syms = subprocess.run(["nm", "-C", ref_cli], capture_output=True, text=True, check=True).stdout
    assert "whb_decoder::decode_02" in syms  # whb.h: the reference's per-type parsers (the mirror has one payload())
    assert "tfrec_amd_submit_host" in syms and "sinked_decoder<tfa1_decoder>::execute_handler" in syms


def test_byte_level_telegrams_through_the_real_decoders(ref_cli, golden_dir, tmp_path):
    cases = json.load(open(os.path.join(golden_dir, "kat_bytes.json")))["cases"]
    for c in cases:
        f = tmp_path / "kat.txt"
        f.write_text(c["hex"] + "\n")
        out = subprocess.run([ref_cli, "-T", "%x" % c["types"], "-X", str(f)], capture_output=True, text=True, check=True).stdout
        assert [ln for ln in out.splitlines() if ln.strip()] == c["text"], c["hex"]


def test_batched_sink_through_the_real_execute_handler_virtual(ref_cli, tmp_path):
    f = tmp_path / "kat.txt"
    f.write_text("2d d4 65 b0 86 20 23 60 e0 56 97\n4b 2d d4 2b 11 02 11 22 33 44 55 00 10 00 d5 07 f6 c4 3b 95 6c\n")
    sink = tmp_path / "sink.out"
    subprocess.run([ref_cli, "-T", "21", "-q", "-X", str(f), "-E", "cat > %s" % sink], check=True)
    recs = [ln.split() for ln in sink.read_text().splitlines()]
    assert [r[:8] for r in recs if r[1] == "65b0"] == [["0", "65b0", "+22.0", "35", "14", "0", "0", "0"]]
    assert any(len(r[1]) == 13 for r in recs)  # the WHB record: 13-digit id (decoder.cpp:84)
    # the reference's own per-record path is untouched: -e runs system("<handler> <args>") (decoder.cpp:94)
    out = subprocess.run([ref_cli, "-T", "1", "-q", "-X", str(f), "-e", "echo REC"], capture_output=True, text=True, check=True).stdout
    assert [ln.split()[1:8] for ln in out.splitlines() if ln.startswith("REC")] == [["65b0", "+22.0", "35", "14", "0", "0", "0"]]
