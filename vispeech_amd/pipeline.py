"""Several batches in flight on one GPU (round 6).

One ``SynthesizerTrn.infer`` is two halves of very different shape: a phoneme- / frame-rate half of ~120 short launches
that leave most of the chip idle (dependent chains of 5-20 us kernels) and a vocoder that fills it.  Consecutive
requests are independent, so a server keeps N contexts of the same model -- each with its own packed weights and
workspaces -- on N HIP streams and issues request k on context k % N: the frame-rate half of request k + 1 overlaps the
vocoder of request k.  Measured on one MI355X (profiles/r06_final_*): the 64-utterance batch 71.6 -> 70.0 ms per batch,
8 utterances 10.8 -> 9.5 ms, one utterance 3.11 -> 2.09 ms, the 60 s utterance 15.5 -> 13.6 ms (throughput; the latency of one
request is unchanged).  The reference has no counterpart (its app serialises requests behind one lock,
inference_api.py:13, 37); ``bench.py --in-flight N`` times exactly this object.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch


class InFlightPool:
    """``n`` contexts of one model on ``n`` streams, used round-robin.

    ``make_net`` builds one ``SynthesizerTrn`` (weights not loaded); ``load`` loads its weights (``net.load_state_dict(sd)``
    on one process, ``sharding.broadcast_weights`` under a process group).  ``nets[0]`` may be passed in ready-made."""

    def __init__(self, make_net: Callable[[], object], load: Callable[[object], None], n: int = 2,
                 first: Optional[object] = None):
        if n < 1:
            raise ValueError("at least one context")
        self.nets: List[object] = []
        for i in range(n):
            if i == 0 and first is not None:
                self.nets.append(first)
                continue
            m = make_net()
            load(m)
            self.nets.append(m)
        # one context runs on the caller's current stream (no stream switch at all: the single-batch form)
        self.streams: List[Optional[torch.cuda.Stream]] = \
            [None] if n == 1 else [torch.cuda.Stream(self.nets[0].device) for _ in self.nets]
        self._k = 0

    def __len__(self) -> int:
        return len(self.nets)

    def next_slot(self) -> Tuple[object, Optional[torch.cuda.Stream]]:
        i = self._k % len(self.nets)
        self._k += 1
        return self.nets[i], self.streams[i]

    def infer(self, *args, after: Optional[Callable[[tuple], None]] = None, **kwargs):
        """``net.infer(*args, **kwargs)`` on the next context, enqueued on that context's stream.  ``after(result)`` runs
        inside the stream's scope (e.g. the start of a gather that orders itself behind this stream's work).  Returns
        ``(result, done)``: ``done`` is an event recorded behind the call -- wait for it (or synchronise) before the
        result is read on another stream or by the host."""
        net, st = self.next_slot()
        if st is None:
            res = net.infer(*args, **kwargs)
            if after is not None:
                after(res)
            ev = None
            if str(getattr(net, "device", "")).startswith("cuda"):
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(net.device))
            return res, ev
        # inputs produced on the caller's stream must be complete before this context's stream reads them -- and their
        # memory must not be handed to a later allocation on the caller's stream while this one still reads it
        st.wait_stream(torch.cuda.current_stream(net.device))
        for x in list(args) + list(kwargs.values()):
            if torch.is_tensor(x) and x.is_cuda:
                x.record_stream(st)
        with torch.cuda.stream(st):
            res = net.infer(*args, **kwargs)
            if after is not None:
                after(res)
            ev = torch.cuda.Event()
            ev.record(st)
        return res, ev

    def restrict(self, n: int) -> "InFlightPool":
        """A view on the first ``n`` contexts (the same objects): e.g. the one-batch-in-flight figure of a bench run."""
        p = object.__new__(InFlightPool)
        p.nets, p.streams, p._k = self.nets[:n], ([None] if n == 1 else self.streams[:n]), 0
        return p
