"""bench.py's launcher logic that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`--gpus 2` inside a job of another size: an error, never a line with a different n_gpus."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "--gpus 2" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
