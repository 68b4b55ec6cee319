"""bench.py's one-process-per-GPU path on a single-GPU box: two ranks share cuda:0 and talk gloo
(VSP_BENCH_BACKEND=gloo test hook; RCCL refuses two ranks on one device).  Exercises, on the real device,
the self-launch of `python bench.py --gpus 2`, the weight-arena broadcast + adopt + header check, the all-reduce
MAX of the frame count, the waveform gather, and the C4 global-batch sharding."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    env = dict(os.environ, VSP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # plain `python bench.py --gpus 2`: the parent spawns the ranks itself (what the driver runs)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", *extra]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                 # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_two_rank_bench_self_launches_on_one_gpu():
    d = run_bench("--batch", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["utterances_per_gpu"] == 4 and d["config"]["parallelism"] == "shard2" and d["config"]["global_batch"] == 8
    assert "cpu_baseline" not in d                            # timed on rank 0 at N = 1 only
    assert d["roofline"]["launches"] == 57 and d["roofline"]["attention"]["launches"] == 8


def test_c4_global_batch_is_sharded_over_the_ranks():
    """C4: ONE global batch (here 6 utterances instead of 256), rank r takes its shard_range slice."""
    d = run_bench("--workload", "C4", "--batch", "6")
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 6 and d["config"]["utterances_per_gpu"] == 3
    # value counts the valid samples of the WHOLE batch once
    from vispeech_amd.synth import WORKLOADS, synth_batch
    b = synth_batch(**dict(WORKLOADS["C4"], batch=6))
    assert d["config"]["valid_samples_per_step"] == 512 * int(b["frame_lengths"].sum())
    assert d["config"]["padded_frames"] == int(b["frame_lengths"].max())
