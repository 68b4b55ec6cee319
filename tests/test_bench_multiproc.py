"""bench.py's one-process-per-GPU path on a single-GPU box: two ranks share cuda:0 and talk gloo
(VSP_BENCH_BACKEND=gloo test hook; RCCL refuses two ranks on one device).  Exercises, on the real device,
the self-launch of `python bench.py --gpus 2`, the weight-arena broadcast + adopt + header check, the all-reduce
MAX of the frame count, the waveform gather, and the C4 global-batch sharding."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, gpus=2, backend="gloo"):
    env = dict(os.environ, VSP_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # plain `python bench.py --gpus N`: the parent spawns the ranks itself (what the driver runs)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", *extra]
    env.setdefault("VSP_BENCH_PG_TIMEOUT", "120")
    # No retry.  A rank that hangs ends ITSELF after VSP_BENCH_DUMP_AFTER seconds -- every thread's Python stack goes to
    # stderr, exit code non-zero -- and torch.distributed.run takes the other rank and the self-launching parent down with
    # it, so a hang arrives here as a failure that says where it hung, well inside the 300 s below.  (Round 3 saw one
    # timeout in five runs of this test; round 5 looped the same command 30x on one box, tools/loop_two_rank.sh:
    # 30 of 30 clean -- profiles/r05_two_rank_loop.txt.)
    env.setdefault("VSP_BENCH_DUMP_AFTER", "200")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                 # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_two_rank_bench_self_launches_on_one_gpu():
    """The default line answers BASELINE.json's metric literally: ONE batch (here 6 utterances instead of 64) split
    shard_range-wise over the ranks -- global_batch is the single batch, not batch x ranks."""
    d = run_bench("--batch", "6")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["n_ranks_seen"] == 2 and "gloo" in d["collectives"]
    assert d["config"]["global_batch"] == 6 and d["config"]["utterances_per_gpu"] == 3 and d["config"]["parallelism"] == "shard2"
    assert "ONE batch split over 2 GPUs" in d["config"]["workload"]
    from vispeech_amd.synth import WORKLOADS, synth_batch
    b = synth_batch(**dict(WORKLOADS["C3"], batch=6))
    assert d["config"]["valid_samples_per_step"] == 512 * int(b["frame_lengths"].sum())      # the ONE batch, counted once
    assert d["config"]["padded_frames"] == int(b["frame_lengths"].max())                     # global padding (G6)
    assert "cpu_baseline" not in d                            # timed on rank 0 at N = 1 only
    assert d["roofline"]["launches"] == 51 and d["roofline"]["attention"]["launches"] == 8
    assert 0 < d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001
    g = d["gather"]
    assert g["bytes_into_rank0_per_step"] == 4 * 512 * d["config"]["padded_frames"] * 3 and 0.0 <= g["share_of_step"] < 1.0


def test_weak_scaling_form_keeps_a_batch_per_rank():
    d = run_bench("--batch", "4", "--weak")
    assert d["scaling"] == "weak" and d["config"]["utterances_per_gpu"] == 4 and d["config"]["global_batch"] == 8
    assert "per GPU" in d["config"]["workload"]


def test_c4_global_batch_is_sharded_over_the_ranks():
    """C4: ONE global batch (here 6 utterances instead of 256), rank r takes its shard_range slice."""
    d = run_bench("--workload", "C4", "--batch", "6")
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 6 and d["config"]["utterances_per_gpu"] == 3
    # value counts the valid samples of the WHOLE batch once
    from vispeech_amd.synth import WORKLOADS, synth_batch
    b = synth_batch(**dict(WORKLOADS["C4"], batch=6))
    assert d["config"]["valid_samples_per_step"] == 512 * int(b["frame_lengths"].sum())
    assert d["config"]["padded_frames"] == int(b["frame_lengths"].max())


def test_rccl_path_runs_on_hardware_with_one_rank():
    """the nccl (= RCCL) branch of bench.py -- init_process_group("nccl", device_id=...), the broadcast
    of the packed weight arena, the frame-count all-reduce, the side-stream waveform gather with persistent buffers,
    the barriers -- executed on the MI355X.  One rank (this box has one GPU; RCCL refuses two ranks per device),
    launched through the self-launch path: the parent never touches the GPU, the rank is a fresh child process."""
    d = run_bench("--dist", "--batch", "4", "--no-cpu-baseline", gpus=1, backend="nccl")
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and "nccl" in d["collectives"]
    assert d["value"] > 0 and d["config"]["utterances_per_gpu"] == 4 and d["config"]["global_batch"] == 4
    assert "per GPU" in d["config"]["workload"]               # (one GPU: the headline line reads as before)
    assert d["roofline"]["launches"] == 51


# ------------------------------------------------------------------------------------------ two RCCL ranks, when the box has them
# RCCL has never run with more than one rank (no multi-GPU node was available to any round).  These
# variants are collected everywhere and skipped on one-GPU boxes, so that the FIRST node with >= 2 devices exercises the
# RCCL broadcast / all-reduce / gather of the path in `pytest -m gpu` before the driver's scaling bench does.
# (torch.cuda.device_count() does not initialise the GPU; the ranks are fresh child processes of bench.py's self-launch.)
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 MI355X: RCCL refuses two ranks on one device")


def _one_gpu_wave(tmp_path, *extra):
    ref = tmp_path / "one_gpu.npy"
    run_bench(*extra, "--dump-wave", str(ref), "--no-cpu-baseline", gpus=1, backend="nccl")
    return np.load(ref)


@two_gpus
def test_rccl_two_ranks_strong_form_matches_the_one_gpu_run(tmp_path):
    """ONE batch of 6 utterances split over two RCCL ranks: weight-arena broadcast + adopt + header check, the frame-count
    all-reduce MAX, the side-stream gather -- rank 0's gathered waveforms equal the one-GPU run of the same batch."""
    got = tmp_path / "two_ranks.npy"
    d = run_bench("--batch", "6", "--dump-wave", str(got), gpus=2, backend="nccl")
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and "nccl" in d["collectives"] and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 6 and d["config"]["utterances_per_gpu"] == 3
    assert d["gather"]["bytes_into_rank0_per_step"] == 4 * 512 * d["config"]["padded_frames"] * 3
    a, b = np.load(got), _one_gpu_wave(tmp_path, "--batch", "6")
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


@two_gpus
def test_rccl_two_ranks_weak_form(tmp_path):
    got = tmp_path / "weak.npy"
    d = run_bench("--batch", "4", "--weak", "--dump-wave", str(got), gpus=2, backend="nccl")
    assert d["scaling"] == "weak" and d["n_ranks_seen"] == 2 and "nccl" in d["collectives"]
    assert d["config"]["utterances_per_gpu"] == 4 and d["config"]["global_batch"] == 8
    assert d["gather"]["bytes_into_rank0_per_step"] == 4 * 512 * d["config"]["padded_frames"] * 4
    a = np.load(got)
    assert a.shape[0] == 8 and np.isfinite(a).all() and np.abs(a[4:]).max() > 0        # rank 1's batch arrived


@two_gpus
def test_rccl_two_ranks_c4_form_matches_the_one_gpu_run(tmp_path):
    got = tmp_path / "c4.npy"
    d = run_bench("--workload", "C4", "--batch", "6", "--dump-wave", str(got), gpus=2, backend="nccl")
    assert d["scaling"] == "strong" and d["n_ranks_seen"] == 2 and d["config"]["global_batch"] == 6
    a, b = np.load(got), _one_gpu_wave(tmp_path, "--workload", "C4", "--batch", "6")
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_dump_wave_hook_on_one_gpu_over_gloo(tmp_path):
    """The hook the two-rank RCCL tests use, exercised on every box: two gloo ranks on one GPU deliver the same
    waveforms as the one-GPU run of the same batch (global padding, gotcha G6; kernel selection differs with the
    shard size, hence the tolerance instead of bit equality)."""
    got = tmp_path / "gloo.npy"
    d = run_bench("--batch", "6", "--dump-wave", str(got))
    assert d["n_ranks_seen"] == 2
    a, b = np.load(got), _one_gpu_wave(tmp_path, "--batch", "6")
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_batches_in_flight_are_disclosed_and_the_single_batch_figure_rides_along(tmp_path):
    """Round 6: by default two batches are in flight per GPU (two contexts on two streams: the frame-rate half of batch
    k + 1 overlaps the generator of batch k).  The line says so, carries the one-batch-in-flight figure of the same K
    steps beside it, and --in-flight 1 is rounds 1-5's headline form; both deliver the same waveforms."""
    a, b = tmp_path / "two.npy", tmp_path / "one.npy"
    d2 = run_bench("--batch", "4", "--no-cpu-baseline", "--dump-wave", str(a), gpus=1, backend="nccl")
    d1 = run_bench("--batch", "4", "--no-cpu-baseline", "--in-flight", "1", "--dump-wave", str(b), gpus=1, backend="nccl")
    assert d2["config"]["batches_in_flight"] == 2 and d1["config"]["batches_in_flight"] == 1
    sb = d2["single_batch"]
    assert sb and sb["ms_per_step"] > 0 and sb["value"] > 0 and d1["single_batch"] is None
    assert d2["roofline"]["launches"] == 51 == d1["roofline"]["launches"]          # the profiled pass runs one batch at a time
    assert np.array_equal(np.load(a), np.load(b))                                  # same kernels on the same inputs: same bits
