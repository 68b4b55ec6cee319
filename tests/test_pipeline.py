"""vispeech_amd.pipeline.InFlightPool: several batches in flight on one GPU (round 6).  The k-th request runs on context
k % N and that context's stream; results are those of a single context, bit for bit (same kernels, same inputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_requests_alternate_contexts_and_match_the_single_context_run():
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    from vispeech_amd.pipeline import InFlightPool
    from vispeech_amd.schema import ModelDims
    from vispeech_amd.synth import synth_batch, synth_state_dict
    sd = synth_state_dict(ModelDims(), seed=1234, infer_only=True)
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    pool = InFlightPool(lambda: SynthesizerTrn(*args, **kwargs).eval(), lambda m: m.load_state_dict(sd), n=2)
    assert len(pool) == 2 and pool.streams[0] is not pool.streams[1] and pool.nets[0] is not pool.nets[1]
    dev = pool.nets[0].device
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    batches = [synth_batch(3, seed=20 + i, mean_phonemes=10, std_phonemes=2, min_phonemes=6, max_phonemes=14, mean_frames=50,
                           jitter_frames=20) for i in range(4)]
    call = lambda runner, b: runner(t(b["phonemes"]), t(b["lengths"]), sid=t(b["sid"]), noise_scale=0.667, noise=t(b["noise"]),
                                    duration_control=t(b["duration"]), pitch_control=t(b["f0"]), energy_control=t(b["energy"]))
    got = [call(pool.infer, b) for b in batches]            # four requests enqueued back to back: two streams, round-robin
    for (res, done) in got:
        done.synchronize()
    ref = [call(pool.nets[0].infer, b) for b in batches]     # the same requests one at a time on context 0
    torch.cuda.synchronize()
    for (res, _), r in zip(got, ref):
        assert torch.equal(res[0], r[0]) and torch.equal(res[2][0], r[2][0])      # waveform and latent: same bits
    one = pool.restrict(1)
    assert len(one) == 1 and one.nets[0] is pool.nets[0] and one.streams == [None]
    res, done = call(one.infer, batches[0])
    done.synchronize()
    assert torch.equal(res[0], ref[0][0])
    with pytest.raises(ValueError):
        InFlightPool(lambda: None, lambda m: None, n=0)
