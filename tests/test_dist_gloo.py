"""N>1 launch contract of bench.py on CPU: two ranks under torch.distributed.run with the gloo backend.
The data path of the benchmark is replica-parallel (no collective): the only cross-rank traffic is the
barrier and the max-over-ranks / sum-over-ranks of two scalars, which is what this exercises."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_aggregation_over_gloo():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--selftest-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2
    assert d["max_elapsed"] == 2.0  # max over ranks of (1 + rank)
    assert d["units"] == 20.0       # sum over ranks


def test_single_process_selftest():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-dist"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["units"] == 10
