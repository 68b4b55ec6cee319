"""BASELINE.json configurations C3 / C4 at FULL size against the oracle (the bench only timed them in round 1).
The GPU runs the whole batch; the oracle (CPU) re-computes a few of its utterances padded to the batch's global
T_p and T_f -- per-utterance results do not depend on the other utterances (SURVEY gotchas G5 / G6), so those
waveforms must agree.  Also the predictors-on modes of the bench (--controls duration | none) at C3 size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
WAVE_TOL, STAGE_TOL = 1e-4, 1e-5


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def setup():
    from oracle.vispeech_oracle import Oracle
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    from vispeech_amd.schema import ModelDims
    from vispeech_amd.synth import synth_state_dict
    dims = ModelDims()
    sd = synth_state_dict(dims, seed=1234, infer_only=True)
    a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
    net = SynthesizerTrn(*a, **kw).eval()
    net.load_state_dict(sd)
    torch.set_num_threads(16)
    return net, Oracle(sd, dims), dims


def gpu_infer(net, batch, sl, tf, controls="all", noise=None):
    t = lambda x: torch.from_numpy(np.asarray(x)).to(net.device)
    kw = {}
    if controls in ("all", "duration"):
        kw["duration_control"] = t(batch["duration"][sl])
    if controls == "all":
        kw["pitch_control"], kw["energy_control"] = t(batch["f0"][sl]), t(batch["energy"][sl])
    return net.infer(t(batch["phonemes"][sl]), t(batch["lengths"][sl]), sid=t(batch["sid"][sl]), noise_scale=0.667,
                     noise=t(noise), t_f=tf, **kw)


def oracle_infer(oracle, batch, idx, tf, noise, controls="all"):
    kw = {}
    if controls in ("all", "duration"):
        kw["duration_control"] = batch["duration"][idx]
    if controls == "all":
        kw["pitch_control"], kw["energy_control"] = batch["f0"][idx], batch["energy"][idx]
    return oracle.infer(batch["phonemes"][idx], batch["lengths"][idx], batch["sid"][idx], noise=noise[idx], noise_scale=0.667,
                        t_f=tf, **kw)


def test_c3_full_batch_matches_oracle_on_zh_and_ja_utterances(setup):
    """BASELINE config 3: the 64-utterance mixed zh/ja batch, every utterance synthesised on the GPU; two zh and two ja
    utterances (the longest and the shortest of each language) re-computed by the oracle with the batch's padding."""
    net, oracle, dims = setup
    from vispeech_amd.synth import JA_RANGE, workload
    b = workload("C3")
    B, tf = b["phonemes"].shape[0], int(b["frame_lengths"].max())
    assert B == 64 and b["noise"].shape == (64, 192, tf)
    is_ja = np.array([JA_RANGE[0] <= b["phonemes"][i, 0] < JA_RANGE[1] for i in range(B)])
    assert is_ja.sum() == 32                                    # "mixed": alternating languages
    o, x_mask, (z, z_p, m_p, logs_p), *_ = gpu_infer(net, b, slice(0, B), tf, noise=b["noise"])
    assert o.shape == (64, 1, 512 * tf)
    fl = b["frame_lengths"]
    zh, ja = np.where(~is_ja)[0], np.where(is_ja)[0]
    idx = np.array([zh[np.argmax(fl[zh])], zh[np.argmin(fl[zh])], ja[np.argmax(fl[ja])], ja[np.argmin(fl[ja])]])
    ref = oracle_infer(oracle, b, idx, tf, b["noise"])
    np.testing.assert_array_equal(x_mask.cpu().numpy()[idx], ref["x_mask"].numpy())
    assert rel_err(z.cpu().numpy()[idx], ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(m_p.cpu().numpy()[idx], ref["m_p"].numpy()) <= STAGE_TOL
    e = rel_err(o.cpu().numpy()[idx], ref["o"].numpy())
    print("C3 full batch, 2 zh + 2 ja utterances vs oracle:", e)
    assert e <= WAVE_TOL


def test_c2_full_batch_matches_oracle(setup):
    """BASELINE config 2 (batch 16 zh utterances of ~5 s, fp32, one GPU; reference models.py:672-722): the whole batch
    on the GPU, the longest and the shortest utterance re-computed by the oracle under the batch's padding."""
    net, oracle, dims = setup
    from vispeech_amd.synth import JA_RANGE, workload
    b = workload("C2")
    B, tf = b["phonemes"].shape[0], int(b["frame_lengths"].max())
    assert B == 16 and b["noise"].shape == (16, dims.inter_channels, tf)
    assert not any(JA_RANGE[0] <= b["phonemes"][i, 0] < JA_RANGE[1] for i in range(B))      # zh only
    o, x_mask, (z, z_p, m_p, logs_p), *_ = gpu_infer(net, b, slice(0, B), tf, noise=b["noise"])
    assert o.shape == (16, 1, 512 * tf)
    fl = b["frame_lengths"]
    idx = np.array([int(np.argmax(fl)), int(np.argmin(fl))])
    ref = oracle_infer(oracle, b, idx, tf, b["noise"])
    np.testing.assert_array_equal(x_mask.cpu().numpy()[idx], ref["x_mask"].numpy())
    assert rel_err(z.cpu().numpy()[idx], ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(logs_p.cpu().numpy()[idx], ref["logs_p"].numpy()) <= STAGE_TOL
    e = rel_err(o.cpu().numpy()[idx], ref["o"].numpy())
    print("C2 full batch, longest + shortest utterance vs oracle:", e)
    assert e <= WAVE_TOL


def test_mixed_language_medium_batch_matches_oracle_completely(setup):
    net, oracle, dims = setup
    from vispeech_amd.synth import synth_batch
    b = synth_batch(6, seed=3103, languages="mixed", mean_phonemes=14, std_phonemes=4, min_phonemes=6, max_phonemes=22,
                    mean_frames=150, jitter_frames=40)
    tf = int(b["frame_lengths"].max())
    o, x_mask, (z, *_), *_ = gpu_infer(net, b, slice(0, 6), tf, noise=b["noise"])
    ref = oracle_infer(oracle, b, np.arange(6), tf, b["noise"])
    assert rel_err(z.cpu().numpy(), ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(o.cpu().numpy(), ref["o"].numpy()) <= WAVE_TOL


def test_c4_shard_of_the_global_batch_matches_oracle(setup):
    """BASELINE config 4: batch 256 sharded over 8 GPUs.  This box has one: it plays rank 5 of 8 -- its 32
    utterances of the GLOBAL batch, padded to the global frame count -- and two of them are checked against the
    oracle under the same padding (the multi-rank mechanics are covered by the gloo tests and test_bench_multiproc)."""
    net, oracle, dims = setup
    from vispeech_amd.sharding import shard_range
    from vispeech_amd.synth import workload
    b = workload("C4")
    assert b["phonemes"].shape[0] == 256
    tf = int(b["frame_lengths"].max())                           # GLOBAL padding
    lo, hi = shard_range(256, 5, 8)
    assert (lo, hi) == (160, 192)
    o, x_mask, (z, *_), *_ = gpu_infer(net, b, slice(lo, hi), tf, noise=b["noise"][lo:hi])
    assert o.shape == (32, 1, 512 * tf)
    idx = np.array([lo + 3, hi - 1])
    ref = oracle_infer(oracle, b, idx, tf, b["noise"])
    assert rel_err(z.cpu().numpy()[idx - lo], ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(o.cpu().numpy()[idx - lo], ref["o"].numpy()) <= WAVE_TOL


@pytest.mark.parametrize("controls", ["duration", "none"])
def test_predictors_on_at_c3_size(setup, controls):
    """The bench's --controls duration | none modes (reference models.py:681-708: F0 / energy / duration predicted):
    full C3 batch on the GPU, two utterances against the oracle; predicted durations must agree EXACTLY."""
    net, oracle, dims = setup
    from vispeech_amd.synth import workload
    b = workload("C3")
    B = 64
    t = lambda x: torch.from_numpy(np.asarray(x)).to(net.device)
    if controls == "duration":
        tf = int(b["frame_lengths"].max())
        noise = b["noise"]
    else:
        enc = net._engine.encode(t(b["phonemes"]), t(b["lengths"]), t(b["sid"]))
        tf = net._engine.frame_lengths_host(enc["frame_lengths"])[1]
        noise = np.random.Generator(np.random.PCG64(31)).standard_normal((B, dims.inter_channels, tf), dtype=np.float32)
    o, x_mask, (z, *_), duration, f0, energy = gpu_infer(net, b, slice(0, B), tf, controls, noise)
    idx = np.array([0, 1])
    ref = oracle_infer(oracle, b, idx, tf, noise, controls)
    np.testing.assert_array_equal(duration.cpu().numpy().reshape(B, -1)[idx], np.asarray(ref["duration"]).reshape(2, -1))
    assert rel_err(f0.cpu().numpy()[idx], ref["F0"].numpy()) <= STAGE_TOL
    assert rel_err(energy.cpu().numpy()[idx], ref["energy"].numpy()) <= STAGE_TOL
    assert rel_err(z.cpu().numpy()[idx], ref["z"].numpy()) <= STAGE_TOL
    assert rel_err(o.cpu().numpy()[idx], ref["o"].numpy()) <= WAVE_TOL
