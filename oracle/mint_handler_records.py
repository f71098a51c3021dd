#!/usr/bin/env python3
"""Mint tests/golden/handler_records.json: the handler command lines the REAL reference builds (decoder::execute_handler,
decoder.cpp:67-96, through store_data's dedupe :46-65 and the -m 1 summary :98-109) for three synthetic streams.
TEST INFRASTRUCTURE; runs only where /root/reference exists.  The last field (ts = time()) is dropped."""
import json
import re
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tfrec_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    if not os.path.isdir(O.REFERENCE_DIR):
        print("no /root/reference here: nothing to mint")
        return 1
    O.build(force=True)
    c = json.load(open(os.path.join(GOLD, "streams.json")))["cases"][0]
    out = dict(case=dict(seed=c["seed"], stream=c["stream"], n_blocks=c["n_blocks"], proto_mask=c["proto_mask"],
                         noise_q8=c["noise_q8"], types=c["types"], thresh=c["thresh"]), streams=[])
    with tempfile.TemporaryDirectory() as tmp:
        for k in range(3):
            iq = synth.gen_stream(c["seed"], c["stream"] + k, c["n_blocks"], c["proto_mask"], c["noise_q8"])
            p = os.path.join(tmp, "s.iq")
            iq.tofile(p)
            rec = {}
            for mode in (0, 1):
                o = subprocess.run([O.REF_DRIVER, "runh", "%x" % c["types"], str(c["thresh"]), "0", p, str(mode)],
                                   capture_output=True, text=True, check=True).stdout
                # (with dbg = -1 the reference still prints its "#nnn <time> <bytes>" debug header, without a newline: a
                # handler line may follow it on the same line)
                rec["mode%d" % mode] = [" ".join(m.split()[:-1]) for m in re.findall(r"REC ([^\n]*)", o)]
            out["streams"].append(rec)
            print("stream %d: %d records (-m 0), %d (-m 1)" % (k, len(rec["mode0"]), len(rec["mode1"])))
    json.dump(out, open(os.path.join(GOLD, "handler_records.json"), "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
