"""world_size-2 gloo test of the N>1 path: streams shard by index across ranks with no data-path
collective; only scalars (max elapsed, counts) are reduced.  The per-stream work is done here by the CPU
oracle (this box has no GPU) -- what is under test is the sharding/aggregation logic bench.py uses."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch.distributed as dist
from tfrec_amd import shard, synth
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N, NB = 6, 8
a, b = shard.shard_range(rank, world, N)
lines = []
for s in range(a, b):
    o = O.Oracle(0x2F, 500)
    o.process(synth.gen_stream(21, s, NB))
    lines += ["%%d %%s" %% (s, ln) for ln in o.text().splitlines()]
dist.barrier()
total = shard.sum_over_ranks(len(lines))
slowest = shard.max_over_ranks(float(rank + 1))
out = [None] * world
dist.all_gather_object(out, lines)
if rank == 0:
    print(json.dumps(dict(total=total, slowest=slowest, lines=sorted(sum(out, [])))))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_matches_single_process(tmp_path):
    import json
    from oracle import oracle as O
    from tfrec_amd import synth

    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    want = []
    for s in range(6):
        o = O.Oracle(0x2F, 500)
        o.process(synth.gen_stream(21, s, 8))
        want += ["%d %s" % (s, ln) for ln in o.text().splitlines()]
    assert res["lines"] == sorted(want) and res["total"] == len(want) and len(want) > 0
    assert res["slowest"] == 2.0
