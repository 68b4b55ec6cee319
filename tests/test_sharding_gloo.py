"""N > 1 path on CPU: two processes over gloo run the sharding helpers of vispeech_amd/sharding.py
(the same functions bench.py and a multi-GPU deployment call) around a stand-in `net` whose
infer() is the CPU oracle.  Checks: balanced contiguous split, global frame padding (gotcha G6),
weight-arena broadcast protocol, gather order, and that the gathered sharded result equals the
unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vispeech_amd.sharding import shard_range


def test_shard_range_is_a_balanced_partition():
    for n in (1, 2, 7, 64, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


class _OracleNet:
    """Stand-in for SynthesizerTrn on CPU: same infer signature, runs the oracle."""

    def __init__(self, oracle):
        self.oracle = oracle

    def infer(self, phonemes, lengths, sid=None, noise_scale=1, max_len=None, energy_control=None, pitch_control=None,
              duration_control=None, *, noise=None, t_f=None):
        r = self.oracle.infer(phonemes.numpy(), lengths.numpy(), sid.numpy(), noise=noise.numpy(), noise_scale=noise_scale,
                              max_len=max_len, energy_control=energy_control.numpy(), pitch_control=pitch_control.numpy(),
                              duration_control=duration_control.numpy(), t_f=t_f)
        return (r["o"], r["x_mask"], (r["z"], r["z_p"], r["m_p"], r["logs_p"]), duration_control, r["F0"], r["energy"])


class _FakeEngine:
    """Arena protocol of Engine.finalize/adopt without a GPU."""

    def __init__(self, n=1000):
        self.n = n
        self.arena = None

    def set_weights(self, sd):
        self.sd = sd

    def finalize(self):
        self.arena = torch.arange(self.n, dtype=torch.float32) * 0.5
        return self.arena

    def adopt(self):
        self.arena = torch.zeros(self.n, dtype=torch.float32)
        self.committed = False
        return self.arena

    def commit_adopted(self):          # after the broadcast: the real engine checks the arena header here
        assert float(self.arena[2]) == 1.0, "commit before the bytes arrived"
        self.committed = True


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle.vispeech_oracle import Oracle
        from vispeech_amd.schema import ModelDims
        from vispeech_amd.sharding import BatchGatherer, broadcast_weights, gather_batch, global_max, infer_sharded, shard_counts
        from vispeech_amd.synth import synth_batch, synth_state_dict
        # 1. collectives
        assert global_max(10 + 5 * rank, "cpu") == 10 + 5 * (world - 1)
        eng = _FakeEngine()
        arena = broadcast_weights(eng, {"w": 1} if rank == 0 else None, src=0)
        assert torch.equal(arena, torch.arange(1000, dtype=torch.float32) * 0.5)
        assert rank == 0 or eng.committed
        shards = gather_batch(torch.full((2 + (rank == 0), 1, 4), float(rank)), dst=0)
        if rank == 0:
            assert [tuple(s.shape) for s in shards] == [(3, 1, 4), (2, 1, 4)]
            assert all(float(s.mean()) == r for r, s in enumerate(shards))
        else:
            assert shards is None
        # persistent gatherer: sizes from shard_range (no exchange), buffers reused over steps, one gather in flight
        counts = shard_counts(5, world)
        assert counts == [3, 2]
        gth = BatchGatherer(counts, (1, 4), torch.float32, "cpu", dst=0)
        bufs = None
        for step in range(3):
            gth.start(torch.full((counts[rank], 1, 4), float(10 * step + rank)))
            got = gth.wait()
            if rank == 0:
                assert [tuple(s.shape) for s in got] == [(3, 1, 4), (2, 1, 4)]
                assert [float(s.mean()) for s in got] == [10.0 * step, 10.0 * step + 1]
                ptrs = [s.data_ptr() for s in got]
                assert bufs is None or bufs == ptrs          # the same receive buffers every step
                bufs = ptrs
            else:
                assert got is None
        with pytest.raises(ValueError):
            gth.start(torch.zeros(counts[rank] + 1, 1, 4))
        assert gather_batch(torch.zeros(counts[rank], 1, 4), dst=0, counts=counts) is None or rank == 0
        # 2. sharded infer == unsharded infer
        dims = ModelDims()
        sd = synth_state_dict(dims, seed=1234, infer_only=True)
        net = _OracleNet(Oracle(sd, dims))
        b = synth_batch(3, seed=55, mean_phonemes=6, std_phonemes=1, min_phonemes=5, max_phonemes=8, mean_frames=14,
                        jitter_frames=3)
        t = torch.from_numpy
        kw = dict(noise_scale=0.667, duration_control=t(b["duration"]), pitch_control=t(b["f0"]), energy_control=t(b["energy"]))
        full, local = infer_sharded(net, t(b["phonemes"]), t(b["lengths"]), t(b["sid"]), noise=t(b["noise"]), **kw)
        lo, hi = shard_range(3, rank, world)
        assert local[0].shape[0] == hi - lo and local[0].shape[-1] == 512 * int(b["frame_lengths"].max())
        if rank == 0:
            ref = net.infer(t(b["phonemes"]), t(b["lengths"]), sid=t(b["sid"]), noise=t(b["noise"]), **kw)[0]
            err = float((full - ref).abs().max() / ref.abs().max())
            q.put(("ok", tuple(full.shape), err))
        else:
            assert full is None
    except Exception as e:  # pragma: no cover
        q.put(("fail", repr(e), rank))
        raise
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_sharding():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    status, shape, err = q.get(timeout=10)
    assert status == "ok", (shape, err)
    assert shape[0] == 3
    assert err <= 2e-6, err      # per-utterance results do not depend on the shard (global padding)
