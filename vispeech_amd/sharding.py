"""Multi-GPU host logic: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on the
MI355X node; "gloo" in the CPU tests).

The synthesis path shards by utterance with no data-path collective: every operator of
``SynthesizerTrn.infer`` is per-utterance (no BatchNorm, no cross-batch reduction),
provided each shard is padded to the GLOBAL frame count (gotcha G6).  The only exchanges are
  1. once:      broadcast of the packed weight arena from rank 0   (``broadcast_weights``)
  2. per batch: all-reduce MAX of one int64, the frame count       (``global_max``)
  3. per batch: gather of the waveforms (and lengths) on rank 0    (``gather_batch``)
The reference has no counterpart (its only parallelism is DDP for training, train.py:62,105-106).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced split of ``n`` utterances: the first ``n % world`` ranks take one more."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _active() -> bool:
    """A process group exists: the collectives run through it even with ONE rank (``bench.py --dist``: the RCCL
    calls of the path execute on a single-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def _world(group=None) -> int:
    return dist.get_world_size(group) if _active() else 1


def global_max(value: int, device, group=None) -> int:
    """MAX over ranks of one integer (the padded frame count of the global batch)."""
    if not _active():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def broadcast_weights(engine, state_dict=None, src: int = 0, group=None) -> torch.Tensor:
    """Rank ``src`` folds + packs the checkpoint; the packed arena (one flat float tensor whose
    layout depends on the config only) is broadcast and adopted by every other rank."""
    rank = dist.get_rank(group) if _active() else 0
    if rank == src:
        if state_dict is None:
            raise ValueError("the source rank needs the state_dict")
        engine.set_weights(state_dict)
        arena = engine.finalize()
    else:
        arena = engine.adopt()
    if _active():
        dist.broadcast(arena, src=src, group=group)
    if rank != src:
        # the received bytes say what rank 0 packed (posterior encoder or not, configuration hash): check them
        if arena.is_cuda:
            torch.cuda.synchronize(arena.device)
        engine.commit_adopted()
    return arena


def shard_counts(n: int, world: int) -> List[int]:
    """Utterances per rank of ``shard_range``: known on every rank without communication."""
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


class BatchGatherer:
    """The per-batch waveform gather on ``dst`` (exchange 3 above) as a PERSISTENT object.

    Shard sizes come from ``shard_range`` (no size exchange, no host read in the step), the receive buffers on
    ``dst`` are allocated once, and the collective runs on a side stream behind an event recorded on the compute
    stream: ``start(o)`` returns at once, so the next step's kernels overlap the transfer over xGMI; ``wait()``
    makes the current stream wait for it and hands out the shards.  One gather is in flight at a time (``start``
    first waits for the previous one).  On CPU tensors (gloo, the tests) the same calls run without streams."""

    def __init__(self, counts: Sequence[int], sample_shape: Sequence[int], dtype=torch.float32, device="cpu",
                 dst: int = 0, group=None):
        self.world = _world(group)
        if len(counts) != self.world:
            raise ValueError("one shard size per rank")
        self.collective = _active()
        self.rank = dist.get_rank(group) if self.collective else 0
        self.counts = [int(c) for c in counts]
        self.dst, self.group = dst, group
        self.device = torch.device(device)
        self.shape = tuple(int(x) for x in sample_shape)
        bmax = max(self.counts)
        mk = lambda: torch.empty((bmax,) + self.shape, dtype=dtype, device=self.device)
        self.recv = [mk() for _ in range(self.world)] if (self.rank == dst and self.collective) else None
        # equal-size operands for the collective: a short shard is staged in a persistent, zero-tailed buffer
        self.stage = torch.zeros((bmax,) + self.shape, dtype=dtype, device=self.device) \
            if self.collective and self.counts[self.rank] < bmax else None
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" and self.collective else None
        self._work = None
        self._done = None
        self._local = None

    def start(self, o: torch.Tensor) -> None:
        if tuple(o.shape) != (self.counts[self.rank],) + self.shape:
            raise ValueError(f"shard has shape {tuple(o.shape)}, expected {(self.counts[self.rank],) + self.shape}")
        self.wait()
        if not self.collective:
            self._local = o
            return
        if self.side is None:
            send = o.contiguous()
            if self.stage is not None:
                self.stage[: o.shape[0]].copy_(send)
                send = self.stage
            self._work = dist.gather(send, self.recv, dst=self.dst, group=self.group, async_op=True)
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            send = o.contiguous()
            if self.stage is not None:
                self.stage[: o.shape[0]].copy_(send, non_blocking=True)
                send = self.stage
            o.record_stream(self.side)
            self._work = dist.gather(send, self.recv, dst=self.dst, group=self.group, async_op=True)
            self._work.wait()              # orders the side stream (not the host) behind the collective
            self._done = torch.cuda.Event()
            self._done.record(self.side)

    def wait(self) -> Optional[List[torch.Tensor]]:
        """Shards in rank order on ``dst`` (views of the persistent buffers: valid until the next ``start``), None
        elsewhere.  Stream-ordered on the GPU: the host does not block."""
        if not self.collective:
            o, self._local = self._local, None
            return None if o is None else [o]
        if self._work is None:
            return None
        if self.side is None:
            self._work.wait()
        else:
            torch.cuda.current_stream(self.device).wait_event(self._done)
        self._work = None
        if self.rank != self.dst:
            return None
        return [b[:c] for b, c in zip(self.recv, self.counts)]


def gather_batch(o: torch.Tensor, dst: int = 0, group=None, counts: Optional[Sequence[int]] = None) -> Optional[List[torch.Tensor]]:
    """One-shot form: gather per-rank waveform shards ``[b_r, 1, S]`` (same S on every rank: global padding; b_r may
    differ by one) on ``dst``.  ``counts`` = shard sizes per rank (``shard_counts``); without them they are
    exchanged first (one small all_gather and a host read -- keep that out of timed loops: ``BatchGatherer``).
    Returns the list of shards on ``dst`` and None elsewhere."""
    world = _world(group)
    if not _active():
        return [o]
    if counts is None:
        nb = torch.tensor([o.shape[0]], dtype=torch.int64, device=o.device)
        got = [torch.zeros_like(nb) for _ in range(world)]
        dist.all_gather(got, nb, group=group)
        counts = torch.cat(got).tolist()
    g = BatchGatherer(counts, o.shape[1:], o.dtype, o.device, dst, group)
    g.start(o)
    shards = g.wait()
    return None if shards is None else [s.clone() for s in shards]


def infer_sharded(net, phonemes, lengths, sid, *, noise: Optional[torch.Tensor] = None, dst: int = 0, group=None,
                  frame_counts: Optional[Sequence[int]] = None, **infer_kwargs):
    """Run ``net.infer`` on this rank's slice of a global batch and gather the waveform on ``dst``.

    Every rank passes the SAME global inputs (phonemes [B,Tp], lengths, sid, control tensors in
    ``infer_kwargs``, noise [B,C,Tf_global]); tensors are sliced along the batch axis here.
    ``frame_counts`` (optional, host ints per utterance) lets ranks agree on the global frame
    count without communication; otherwise it is the all-reduce MAX of the local maxima that
    ``net.infer`` reports.  With ``noise_seed`` in ``infer_kwargs`` instead of ``noise`` the library draws the noise: this
    rank then draws ITS utterances' part of the global [B, C, Tf] stream (``noise_offset`` = lo * C * Tf), so the
    result does not depend on the shard layout.  Returns (o_full [B,1,S] on ``dst`` else None, local result tuple)."""
    world = _world(group)
    rank = dist.get_rank(group) if _active() else 0
    B = phonemes.shape[0]
    lo, hi = shard_range(B, rank, world)
    sl = slice(lo, hi)
    kw = {}
    for k, v in infer_kwargs.items():
        kw[k] = v[sl] if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == B else v
    if frame_counts is not None:
        t_f = int(max(frame_counts))
    elif noise is not None:
        t_f = int(noise.shape[-1])
    else:
        raise ValueError("pass frame_counts or noise so that all ranks pad to the same frame count")
    t_f = global_max(t_f, phonemes.device, group)
    if noise is None and kw.get("noise_seed") is not None:
        kw["noise_offset"] = lo * int(net.dims.inter_channels) * t_f
    out = net.infer(phonemes[sl], lengths[sl], sid=sid[sl], noise=None if noise is None else noise[sl], t_f=t_f, **kw)
    shards = gather_batch(out[0], dst=dst, group=group, counts=shard_counts(B, world))
    full = torch.cat(shards, dim=0) if shards is not None else None
    return full, out
