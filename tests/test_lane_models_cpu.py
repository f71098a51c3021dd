"""CPU checks of the two lane-per-step constructions of the cooperative slicers: the Python
models in profiles/ubench/ restate what coop_tfa1 / coop_tfa2's group_vec do per lane and compare, on adversarial random
candidate strings, with the sample-by-sample rules of tfa1.cpp:164-177 / tfa2.cpp:383-411 + decoder.cpp:118-122 (block-relative
last_bit_idx, the rebase with its == 0 sentinel).  The GPU tests compare the kernels themselves with the oracle; these pin the
arguments the kernels rest on (history-free runs + parity carry; speculate / compare / re-walk) without a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,trials", [("tfa1_lane_model.py", 11, 80), ("tfa1_lane_model.py", 12, 80),
                                                ("tfa2_lane_model.py", 11, 25), ("tfa2_lane_model.py", 12, 25)])
def test_lane_per_step_model_equals_the_sample_by_sample_rules(script, seed, trials):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "ubench", script), str(seed), str(trials)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.startswith("ok: groups vector")
    # both paths of the model were exercised: groups done lane-parallel and groups left to the scalar walk
    words = p.stdout.replace(",", " ").split()
    assert int(words[3]) > 0 and int(words[5]) > 0
