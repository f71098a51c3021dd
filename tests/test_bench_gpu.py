"""bench.py itself, on the GPU box: the single-rank line and the N>1 code path (two ranks of bench.py on ONE device, gloo
for the barrier / scalar reduces -- the data path has no collective), so that the rendezvous, the barrier + MAX-reduce
bracket and the per-rank seeding have run before the driver's 8-GPU launch."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    return json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])


COMMON = ["--steps", "3", "--warmup", "1", "--streams", "64", "--blocks", "8", "--cpu-budget", "0"]


def test_bench_single_rank_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + ["--h2d-steps", "2"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "MSamples/s"
    c = j["config"]
    assert c["parity_ok"] is True and c["parity_gate_streams"] == 64  # every stream of the batch
    assert c["atan_host_mismatch"] == 0
    # slow-path decisions beyond the per-submit log are exact by construction but not compared with this host's libm: none
    # unless a submit resolved more than the log holds
    assert c["atan_unverified"] == 0 or c["atan_resolved"] > 62
    # the batch AFTER the timed region (carried state, FIFO four deep) against the oracle continued over every repetition
    assert c["parity_after_timed"] is True and c["parity_after_timed_streams"] == 64
    assert c["parity_after_timed_batches_carried"] == 1 + 1 + 3
    rf = j["roofline"]
    assert rf["hbm_floor_ms"] > 0 and rf["algorithmic_valu_floor_ms"] > rf["hbm_floor_ms"]
    assert j["ms_min"] <= j["ms_median"] <= j["ms_max"]
    assert 0 < j["roofline"]["frac"] < 1 and 0 < j["roofline"]["whole_path_frac"] < 1
    assert j["h2d_included"]["value"] > 0 and j["h2d_included"]["value"] <= j["value"] * 1.05


def test_bench_two_ranks_gloo_on_one_device():
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                                       "--same-device"] + COMMON, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith('{"metric"')]  # only rank 0 prints
    j = _line(outs[0][0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    c = j["config"]
    assert c["parity_ok"] is True and c["dist_backend"] == "gloo" and c["atan_host_mismatch"] == 0
    assert len(c["rank_input_crc32"]) == 2 and c["rank_input_crc32"][0] != c["rank_input_crc32"][1]  # distinct streams per rank
    assert c["events_all_ranks"] > c["events_per_step"] * j["steps"]  # both ranks' events counted
    # value is the whole job: both ranks' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 64 * 8 * 32768 * j["steps"] / (j["ms_per_step"] * j["steps"] * 1e-3) / 1e6) / j["value"] < 1e-3


def test_bench_gpus_2_launches_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no RANK / WORLD_SIZE in the environment): bench.py starts the two ranks itself
    through torch.distributed.run and reports n_gpus = the ranks that really ran (it used to run ONE rank and
    print n_gpus 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--dist-backend",
                          "gloo"] + COMMON, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parity_ok"] is True
    assert len(j["config"]["rank_input_crc32"]) == 2


def test_bench_gpus_8_same_device():
    """BASELINE configs[3] dry run: EIGHT ranks of bench.py (64 streams x 8 blocks each) on one device, gloo for the barrier and
    the scalar reduces -- rendezvous of 8, per-rank seeding, the barrier + MAX-reduce bracket, 8 contexts' memory and the
    whole-job value at N = 8, before an 8-GPU node ever runs it.  No scaling figure is taken from this."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--same-device", "--dist-backend",
                          "gloo"] + COMMON, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["value"] > 0
    c = j["config"]
    assert c["parity_ok"] is True and c["parity_after_timed"] is True and c["atan_host_mismatch"] == 0
    assert len(set(c["rank_input_crc32"])) == 8  # eight distinct sets of streams
    assert c["events_all_ranks"] > 4 * c["events_per_step"] * j["steps"]
    assert abs(j["value"] - 8 * 64 * 8 * 32768 * j["steps"] / (j["ms_per_step"] * j["steps"] * 1e-3) / 1e6) / j["value"] < 1e-3
