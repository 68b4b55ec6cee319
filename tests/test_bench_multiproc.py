"""bench.py's one-process-per-GPU path on a single-GPU box: two ranks share cuda:0 and talk gloo
(VSP_BENCH_BACKEND=gloo test hook; RCCL refuses two ranks on one device).  Exercises, on the real device,
the weight-arena broadcast + adopt, the all-reduce MAX of the frame count and the waveform gather."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_on_one_gpu():
    env = dict(os.environ, VSP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                 # rank 0 prints ONE json line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["utterances_per_gpu"] == 4 and d["config"]["parallelism"] == "shard2"
    assert "cpu_baseline" not in d                            # timed on rank 0 at N = 1 only
