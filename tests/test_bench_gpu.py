"""bench.py on a GPU box: the one-line contract at N = 1, and two ranks sharing the GPU through the slice-sharded C++ host
(gloo carries the collectives here; the driver's multi-GPU runs use RCCL through the same callbacks)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(text[-2000:])


@pytest.mark.gpu
def test_bench_line_and_two_ranks_on_one_gpu():
    one = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in a, key
    assert a["n_gpus"] == 1 and a["steps"] == 3 and a["value"] > 0 and a["dtype"] == "f32" and a["roofline"]["frac"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", "bench.py", "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1",
                          "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["value"] > 0
    # weak scaling: the tiny workload is not multiplied (only P4 / S8 are); both ranks together cover the same active pixels
    assert b["config"]["Va_total"] == a["config"]["Va_total"] and b["config"]["slices"] == a["config"]["slices"]
    assert 0 < b["config"]["Va_rank0"] < b["config"]["Va_total"]
