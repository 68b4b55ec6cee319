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


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def global_max(value: int, device, group=None) -> int:
    """MAX over ranks of one integer (the padded frame count of the global batch)."""
    if _world(group) == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def broadcast_weights(engine, state_dict=None, src: int = 0, group=None) -> torch.Tensor:
    """Rank ``src`` folds + packs the checkpoint; the packed arena (one flat float tensor whose
    layout depends on the config only) is broadcast and adopted by every other rank."""
    rank = dist.get_rank(group) if _world(group) > 1 else 0
    if rank == src:
        if state_dict is None:
            raise ValueError("the source rank needs the state_dict")
        engine.set_weights(state_dict)
        arena = engine.finalize()
    else:
        arena = engine.adopt()
    if _world(group) > 1:
        dist.broadcast(arena, src=src, group=group)
    if rank != src:
        # the received bytes say what rank 0 packed (posterior encoder or not, configuration hash): check them
        if arena.is_cuda:
            torch.cuda.synchronize(arena.device)
        engine.commit_adopted()
    return arena


def gather_batch(o: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather per-rank waveform shards ``[b_r, 1, S]`` (same S on every rank: global padding; b_r may
    differ by one) on ``dst``.  Returns the list of shards on ``dst`` and None elsewhere."""
    world = _world(group)
    if world == 1:
        return [o]
    rank = dist.get_rank(group)
    nb = torch.tensor([o.shape[0]], dtype=torch.int64, device=o.device)
    counts = [torch.zeros_like(nb) for _ in range(world)]
    dist.all_gather(counts, nb, group=group)
    bmax = int(max(int(c.item()) for c in counts))
    if o.shape[0] < bmax:                      # equal-size buffers for the collective
        pad = torch.zeros((bmax - o.shape[0],) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
        o = torch.cat([o, pad], dim=0)
    o = o.contiguous()
    bufs = [torch.empty_like(o) for _ in range(world)] if rank == dst else None
    dist.gather(o, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[: int(c.item())] for b, c in zip(bufs, counts)]


def infer_sharded(net, phonemes, lengths, sid, *, noise: Optional[torch.Tensor] = None, dst: int = 0, group=None,
                  frame_counts: Optional[Sequence[int]] = None, **infer_kwargs):
    """Run ``net.infer`` on this rank's slice of a global batch and gather the waveform on ``dst``.

    Every rank passes the SAME global inputs (phonemes [B,Tp], lengths, sid, control tensors in
    ``infer_kwargs``, noise [B,C,Tf_global]); tensors are sliced along the batch axis here.
    ``frame_counts`` (optional, host ints per utterance) lets ranks agree on the global frame
    count without communication; otherwise it is the all-reduce MAX of the local maxima that
    ``net.infer`` reports.  Returns (o_full [B,1,S] on ``dst`` else None, local result tuple)."""
    world = _world(group)
    rank = dist.get_rank(group) if world > 1 else 0
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
    out = net.infer(phonemes[sl], lengths[sl], sid=sid[sl], noise=None if noise is None else noise[sl], t_f=t_f, **kw)
    shards = gather_batch(out[0], dst=dst, group=group)
    full = torch.cat(shards, dim=0) if shards is not None else None
    return full, out
