"""The callers either side of the path: checkpoint loader, phoneme-ID front door,
service wrapper / streamed output.  CPU tests cover the host logic; `-m gpu` tests drive real files and rows
through the C-ABI and compare with goldens produced by the real reference."""
import io
import os
import threading
import wave

import numpy as np
import pytest
import torch

WAVE_TOL, STAGE_TOL = 1e-4, 1e-5


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ------------------------------------------------------------------------------------------ CPU: host logic
def test_merge_checkpoint_state_is_tolerant_like_the_reference():
    """reference utils.py:32-43: missing keys and wrong shapes keep the model's value and are reported."""
    from vispeech_amd.utils import merge_checkpoint_state
    schema = {"a.weight": (2, 3), "b.bias": (4,), "c.weight": (1,), "d.gone": (2,)}
    current = {k: np.full(s, 7.0, np.float32) for k, s in schema.items() if k != "d.gone"}
    saved = {"a.weight": torch.ones(2, 3), "b.bias": torch.ones(5), "extra.key": torch.ones(1)}
    new, missing, mismatched = merge_checkpoint_state(saved, schema, current)
    assert missing == ["c.weight", "d.gone"] and mismatched == ["b.bias"]
    assert torch.equal(new["a.weight"], torch.ones(2, 3))            # taken from the checkpoint
    assert (new["b.bias"] == 7.0).all() and (new["c.weight"] == 7.0).all()   # kept
    assert "d.gone" not in new and "extra.key" not in new            # nothing to keep / not a model key


def reference_style_load(checkpoint_path, model, optimizer=None):
    """What the reference's OWN loader does (reference utils.py:21-51), restated here step for step -- it is the code a
    user of inference.py:36 runs against whatever class `SynthesizerTrn` names: read the file, walk the keys of the
    MODEL's state_dict() (taken before anything was loaded), take the checkpoint's tensor where it is there with the
    model's shape, else keep the model's value, then a strict load_state_dict of the merged dict."""
    assert os.path.isfile(checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    saved = ckpt["model"]
    target = model.module if hasattr(model, "module") else model
    merged, kept = {}, []
    for key, val in target.state_dict().items():
        try:
            merged[key] = saved[key]
            assert saved[key].shape == val.shape, (saved[key].shape, val.shape)
        except Exception:
            kept.append(key)
            merged[key] = val
    target.load_state_dict(merged)
    return model, optimizer, ckpt["learning_rate"], ckpt["iteration"], kept


class _RecordingEngine:
    """Stands in for the GPU engine in the CPU test of the shim's state_dict surface."""
    def __init__(self, dims, device="cuda:0"):
        self.dims, self.device, self.ready, self.loaded = dims, torch.device(device), False, None

    def set_weights(self, state_dict, strict=True):
        self.loaded = dict(state_dict)
        return [], []

    def finalize(self):
        self.ready = True


def test_the_references_loader_runs_against_the_shim_as_written(tmp_path, monkeypatch):
    """VERDICT r5 missing #3: reference utils.py:29-46 iterates `model.state_dict()` BEFORE the first load; the shim
    must hand it the 753 schema keys with the right shapes (zeros), so that the merged dict is the checkpoint and the
    strict load passes.  Host logic only: the engine is a recorder."""
    import vispeech_amd.models as vm
    from vispeech_amd import config as vcfg
    from vispeech_amd.schema import ModelDims, state_dict_schema
    from vispeech_amd.synth import synth_state_dict
    monkeypatch.setattr(vm, "Engine", _RecordingEngine)
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    net = vm.SynthesizerTrn(*args, **kwargs).eval()
    schema = state_dict_schema(ModelDims())
    before = net.state_dict()
    assert list(before) == list(schema) and len(before) == 753
    assert all(tuple(before[k].shape) == tuple(schema[k]) and before[k].dtype == torch.float32 for k in schema)
    assert not hasattr(net, "module")                        # the loader's hasattr(model, 'module') branch: plain model
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(ModelDims(), seed=1234).items()}
    p = tmp_path / "G_76000.pth"
    torch.save({"model": sd, "iteration": 76000, "optimizer": {}, "learning_rate": 2e-4}, str(p))
    m, _, lr, it, kept = reference_style_load(str(p), net, None)
    assert m is net and (lr, it) == (2e-4, 76000) and kept == []
    assert set(net._engine.loaded) == set(schema) and net._engine.ready
    after = net.state_dict()
    assert all(torch.equal(after[k], sd[k]) for k in schema)
    # a checkpoint without one key and with one foreign shape: the loader keeps the model's (loaded) values for both
    sd2 = dict(sd)
    del sd2["dec.conv_post.weight"]
    sd2["emb_g.weight"] = torch.zeros(3, 5)
    p2 = tmp_path / "G_77000.pth"
    torch.save({"model": sd2, "iteration": 77000, "optimizer": {}, "learning_rate": 1e-4}, str(p2))
    *_, kept2 = reference_style_load(str(p2), net, None)
    assert kept2 == ["dec.conv_post.weight", "emb_g.weight"] or set(kept2) == {"dec.conv_post.weight", "emb_g.weight"}
    assert torch.equal(net.state_dict()["emb_g.weight"], sd["emb_g.weight"])


class _FakeNet:
    """Stands in for SynthesizerTrn in the lock tests: infer blocks until released."""
    class dims:
        total_upsample = 4
        inter_channels = 2
    device = "cpu"

    def __init__(self):
        self.started, self.go = threading.Event(), threading.Event()

    def infer(self, ph, ln, **kw):
        self.started.set()
        assert self.go.wait(10)
        B = ph.shape[0]
        o = torch.linspace(-1, 1, B * 12).reshape(B, 1, 12)
        x_mask = torch.ones(B, 1, 3, dtype=torch.bool)
        x_mask[0, 0, 2] = False
        return (o, x_mask, None, None, None, None)


def test_a_stream_that_is_never_started_does_not_keep_the_lock():
    """stream() takes the single-flight lock before the first chunk exists; a caller that drops or closes
    the stream without ever advancing it must still release it (a bare generator would never run its finally)."""
    import gc
    from vispeech_amd.service import Busy, SynthesisService
    svc = SynthesisService(_FakeNet())
    batch = dict(phonemes=np.zeros((2, 3), np.int64), lengths=np.array([3, 3]), sid=np.array([0, 1]))
    it = svc.stream(batch)
    assert svc.busy
    with pytest.raises(Busy):
        svc.stream(batch)
    del it                                   # never advanced
    gc.collect()
    assert not svc.busy
    it = svc.stream(batch)
    it.close()
    it.close()                               # idempotent
    assert not svc.busy
    assert list(it) == []                    # a closed stream yields nothing and does not touch the lock again
    with svc.stream(batch) as it2:
        assert svc.busy
    assert not svc.busy
    # a stream whose first chunk fails (the fake net has no engine) releases the lock too
    it = svc.stream(batch)
    with pytest.raises(AttributeError):
        next(it)
    assert not svc.busy


def test_service_is_single_flight_and_never_queues():
    """reference inference_api.py:13, 37: mutex.acquire(blocking=False) -- a second request is refused at once."""
    from vispeech_amd.service import Busy, SynthesisService, pcm16
    net = _FakeNet()
    svc = SynthesisService(net)
    batch = dict(phonemes=np.zeros((2, 3), np.int64), lengths=np.array([3, 3]), sid=np.array([0, 1]))
    res = {}
    th = threading.Thread(target=lambda: res.update(first=svc.synthesize(batch)))
    th.start()
    assert net.started.wait(10) and svc.busy
    assert svc.synthesize(batch) is None                 # refused, not queued
    assert svc.wav_bytes(batch) is None
    with pytest.raises(Busy):
        svc.stream(batch)
    net.go.set()
    th.join(10)
    assert not svc.busy
    first = res["first"]
    assert first.dtype == np.dtype("<i2") and first.size == 2 * 4          # valid frames only (mask), hop 4
    np.testing.assert_array_equal(first, pcm16(torch.linspace(-1, 1, 24)[:8]))
    wav = svc.wav_bytes(batch, utterance=1)              # the lock was released: served again
    with wave.open(io.BytesIO(wav), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 44100, 12)


def test_in_flight_pool_host_logic_with_one_context():
    """vispeech_amd.pipeline.InFlightPool without a GPU: one context = the caller's stream, no event; the loader runs once
    per context that the pool builds itself; `after` sees the result; n < 1 is refused."""
    from vispeech_amd.pipeline import InFlightPool
    built, loaded, seen = [], [], []

    class Net:
        device = "cpu"

        def infer(self, x, k=0):
            return (x + k,)

    def make():
        built.append(1)
        return Net()

    pool = InFlightPool(make, lambda m: loaded.append(m), n=1)
    assert len(pool) == 1 and pool.streams == [None] and len(built) == 1 and loaded == pool.nets
    res, ev = pool.infer(2, k=3, after=lambda r: seen.append(r))
    assert res == (5,) and ev is None and seen == [(5,)]
    first = Net()
    pool2 = InFlightPool(make, lambda m: loaded.append(m), n=1, first=first)      # a ready-made first context is adopted as is
    assert pool2.nets == [first] and len(built) == 1
    assert pool.restrict(1).nets == pool.nets
    with pytest.raises(ValueError):
        InFlightPool(make, lambda m: None, n=0)


def test_pooled_service_serves_n_requests_and_refuses_the_next():
    """Round 6: N single-flight slots (one per context of an InFlightPool) -- the reference's "refuse, never queue" with N
    locks instead of one (inference_api.py:13, 37): two requests run side by side, the third is refused at once, a freed
    slot serves again."""
    from types import SimpleNamespace
    from vispeech_amd.service import Busy, PooledSynthesisService, pcm16
    nets = [_FakeNet(), _FakeNet()]
    svc = PooledSynthesisService(SimpleNamespace(nets=nets, streams=[None]))
    batch = dict(phonemes=np.zeros((2, 3), np.int64), lengths=np.array([3, 3]), sid=np.array([0, 1]))
    res = {}
    th = [threading.Thread(target=lambda k=k: res.update({k: svc.synthesize(batch)})) for k in range(2)]
    th[0].start()
    assert nets[0].started.wait(10) and not svc.busy          # slot 0 taken, slot 1 free
    th[1].start()
    assert nets[1].started.wait(10) and svc.busy              # both taken
    assert svc.synthesize(batch) is None and svc.wav_bytes(batch) is None
    with pytest.raises(Busy):
        svc.stream(batch)
    nets[1].go.set()
    th[1].join(10)
    assert not svc.busy                                        # slot 1 is free again while slot 0 still runs
    assert svc.synthesize(batch) is not None                   # ... and serves (its net no longer blocks)
    nets[0].go.set()
    th[0].join(10)
    for k in range(2):
        np.testing.assert_array_equal(res[k], pcm16(torch.linspace(-1, 1, 24)[:8]))


# ------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dims():
    from vispeech_amd.schema import ModelDims
    return ModelDims()


@pytest.fixture(scope="module")
def full_weights(dims):
    from vispeech_amd.synth import synth_state_dict
    return synth_state_dict(dims, seed=1234)          # the full 753-tensor schema, as a reference checkpoint holds


def make_net():
    from vispeech_amd import config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    args, kwargs = vcfg.synthesizer_args(vcfg.default_hparams())
    return SynthesizerTrn(*args, **kwargs).eval()


def golden_infer(net, g):
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    return net.infer(t(g["in_phonemes"]), t(g["in_lengths"]), sid=t(g["in_sid"]), noise_scale=float(g["in_noise_scale"]),
                     duration_control=t(g["in_duration"]), pitch_control=t(g["in_f0"]), energy_control=t(g["in_energy"]),
                     noise=t(g["in_noise"]))


@gpu
def test_checkpoint_file_loads_and_matches_reference_golden(tmp_path, dims, full_weights, golden_dir):
    """a file in the reference's save format (utils.py:67-70) goes through
    utils.load_checkpoint into the GPU path and reproduces the reference's outputs; then a second file with a
    missing key and a wrong-shape key is loaded ON TOP (tolerant branches, utils.py:34-43): nothing changes."""
    from vispeech_amd.utils import load_checkpoint
    sd = {k: torch.from_numpy(v) for k, v in full_weights.items()}
    opt_state = {"state": {0: {"step": torch.tensor(12.0), "exp_avg": torch.zeros(3)}}, "param_groups": [{"lr": 2e-4, "params": [0]}]}
    p = tmp_path / "G_1000.pth"
    torch.save({"model": sd, "iteration": 1000, "optimizer": opt_state, "learning_rate": 2e-4}, str(p))
    net = make_net()

    class Opt:
        loaded = None

        def load_state_dict(self, s):
            Opt.loaded = s

    opt = Opt()
    m, o2, lr, it = load_checkpoint(str(p), net, opt)
    assert m is net and o2 is opt and lr == 2e-4 and it == 1000 and Opt.loaded["param_groups"][0]["lr"] == 2e-4
    g = np.load(os.path.join(golden_dir, "ragged_controls.npz"))
    o, x_mask, (z, z_p, m_p, logs_p), *_ = golden_infer(net, g)
    assert rel_err(o.cpu().numpy(), g["o"]) <= WAVE_TOL and rel_err(z.cpu().numpy(), g["z"]) <= STAGE_TOL
    # tolerant reload: one key missing, one with a foreign shape, one changed
    sd2 = dict(sd)
    del sd2["dec.conv_post.weight"]
    sd2["emb_g.weight"] = torch.zeros(3, 5)
    p2 = tmp_path / "G_2000.pth"
    torch.save({"model": sd2, "iteration": 2000, "optimizer": opt_state, "learning_rate": 1e-4}, str(p2))
    _, _, lr2, it2 = load_checkpoint(str(p2), net, None)
    assert (lr2, it2) == (1e-4, 2000)
    o2_, *_ = golden_infer(net, g)
    assert torch.equal(o2_, o)                                       # kept values: bit-identical outputs
    # a pickle that needs more than the restricted unpickler is refused unless the caller opts in
    p3 = tmp_path / "G_bad.pth"
    torch.save({"model": sd, "iteration": 1, "optimizer": opt_state, "learning_rate": 1e-4, "hook": threading.Lock}, str(p3))
    with pytest.raises(Exception):
        load_checkpoint(str(p3), net, None)


@gpu
def test_the_references_loader_as_written_reaches_the_golden(tmp_path, dims, full_weights, golden_dir):
    """inference.py:26-44 as a user runs it: construct, eval(), the reference's utils.load_checkpoint (restated above:
    `reference_style_load`) on a reference-format file, infer -- against the reference's golden outputs."""
    sd = {k: torch.from_numpy(v) for k, v in full_weights.items()}
    p = tmp_path / "G_76000.pth"
    torch.save({"model": sd, "iteration": 76000, "optimizer": {}, "learning_rate": 2e-4}, str(p))
    net = make_net()
    _ = net.eval()
    _, _, lr, it, kept = reference_style_load(str(p), net, None)
    assert (lr, it) == (2e-4, 76000) and kept == []
    g = np.load(os.path.join(golden_dir, "ragged_controls.npz"))
    o, x_mask, (z, *_), *_ = golden_infer(net, g)
    assert rel_err(o.cpu().numpy(), g["o"]) <= WAVE_TOL and rel_err(z.cpu().numpy(), g["z"]) <= STAGE_TOL


@gpu
def test_val_filelist_rows_through_the_front_door(dims, full_weights, golden_dir):
    """rows of the reference's filelists/val.list -> parse_filelist_row -> collate_rows ->
    infer on the GPU, against the reference's own front door + infer (golden made by make_golden.py vallist)."""
    from vispeech_amd.text import SymbolTable, collate_rows, parse_filelist_row
    g = np.load(os.path.join(golden_dir, "val_filelist.npz"))
    table = SymbolTable([str(s) for s in g["symbols"]])
    spk2id = {str(k): int(v) for k, v in zip(g["spk2id_keys"], g["spk2id_vals"])}
    rows = [parse_filelist_row(str(l)) for l in g["rows"]]
    batch = collate_rows(rows, table, spk2id)
    for k, gk in (("phonemes", "in_phonemes"), ("lengths", "in_lengths"), ("sid", "in_sid"), ("duration", "in_duration"),
                  ("f0", "in_f0"), ("energy", "in_energy")):
        np.testing.assert_array_equal(batch[k], g[gk])               # same ids / controls as the reference's parser
    net = make_net()
    net.load_state_dict(full_weights)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(net.device)
    o, x_mask, (z, z_p, m_p, logs_p), *_ = net.infer(
        t(batch["phonemes"]), t(batch["lengths"]), sid=t(batch["sid"]), noise_scale=float(g["in_noise_scale"]),
        duration_control=t(batch["duration"]), pitch_control=t(batch["f0"]), energy_control=t(batch["energy"]),
        noise=t(g["in_noise"]))
    np.testing.assert_array_equal(x_mask.cpu().numpy(), g["x_mask"])
    for name, val in (("m_p", m_p), ("logs_p", logs_p), ("z", z)):
        assert rel_err(val.cpu().numpy(), g[name]) <= STAGE_TOL, name
    assert rel_err(o.cpu().numpy(), g["o"]) <= WAVE_TOL


@gpu
def test_service_streams_the_same_bytes_and_refuses_while_streaming(dims, full_weights, golden_dir):
    """chunked PCM16 streaming over the halo streamer == the one-shot waveform, byte for byte;
    the single-flight lock is held while a stream is open."""
    from vispeech_amd.service import Busy, SynthesisService
    g = np.load(os.path.join(golden_dir, "val_filelist.npz"))
    net = make_net()
    net.load_state_dict(full_weights)
    svc = SynthesisService(net, chunk_frames=100)
    batch = dict(phonemes=g["in_phonemes"], lengths=g["in_lengths"], sid=g["in_sid"], duration=g["in_duration"],
                 f0=g["in_f0"], energy=g["in_energy"])
    noise = torch.from_numpy(g["in_noise"]).to(net.device)
    for utt in (0, 1):
        whole = svc.synthesize(batch, utterance=utt, noise=noise)
        frames = int(g["in_duration"][utt].sum())
        assert whole.size == frames * 512
        it = svc.stream(batch, utterance=utt, noise=noise)
        first = next(it)
        assert len(first) == 2 * 100 * 512 and svc.busy
        assert svc.synthesize(batch) is None                         # refused while the stream is open
        with pytest.raises(Busy):
            svc.stream(batch)
        rest = b"".join(it)
        assert not svc.busy
        assert first + rest == whole.tobytes()
    # the reference's golden waveform, as PCM16, within one LSB
    from vispeech_amd.service import pcm16
    ref = pcm16(g["o"][0, 0, : int(g["in_duration"][0].sum()) * 512])
    assert np.abs(svc.synthesize(batch, 0, noise).astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # closing a stream early releases the lock
    it = svc.stream(batch, noise=noise)
    next(it)
    it.close()
    assert not svc.busy


@gpu
def test_stream_chunk_entry_point_is_bit_identical_to_the_full_vocoder(dims, full_weights):
    """The C-ABI streamed-generator call (vsp_generator_stream_chunk): any chunking == one vsp_generator call."""
    import ctypes as C
    net = make_net()
    net.load_state_dict(full_weights)
    eng = net._engine
    rng = np.random.Generator(np.random.PCG64(5))
    B, T = 2, 333
    z = torch.from_numpy(rng.standard_normal((B, dims.inter_channels, T), dtype=np.float32)).cuda()
    gv = torch.from_numpy(full_weights["emb_g.weight"][[3, 40]]).cuda()
    full = eng.generator(z, gv)
    halo = eng.generator_halo
    assert 13 <= halo <= 16
    lib = eng.lib
    ws_bytes = lib.vsp_generator_stream_workspace_bytes(eng.ctx, B, 120)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for f0, f1 in ((0, 120), (120, 121), (121, 240), (240, 333), (5, 37)):
        o = torch.empty(B, 1, (f1 - f0) * 512, device="cuda")
        rc = lib.vsp_generator_stream_chunk(eng.ctx, stream, B, T, C.c_void_p(z.data_ptr()), C.c_void_p(gv.data_ptr()),
                                            f0, f1, C.c_void_p(o.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes)
        assert rc == 0, lib.vsp_last_error(eng.ctx)
        torch.cuda.synchronize()
        assert torch.equal(o, full[:, :, f0 * 512:f1 * 512]), (f0, f1)
    # error paths: bad range, short workspace
    o = torch.empty(B, 1, 512, device="cuda")
    assert lib.vsp_generator_stream_chunk(eng.ctx, stream, B, T, C.c_void_p(z.data_ptr()), C.c_void_p(gv.data_ptr()),
                                          10, 10, C.c_void_p(o.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes) == -1
    assert lib.vsp_generator_stream_chunk(eng.ctx, stream, B, T, C.c_void_p(z.data_ptr()), C.c_void_p(gv.data_ptr()),
                                          0, 120, C.c_void_p(o.data_ptr()), C.c_void_p(ws.data_ptr()), 1024) == -6


@gpu
def test_typed_and_device_weights_equal_the_float32_host_load(dims, full_weights, golden_dir):
    """vsp_set_weight_typed (dtype + host-or-device pointer): a bf16 / fp16 / fp64 checkpoint, on the
    host or on the device, gives bit-for-bit what its float32 up-cast gives."""
    g = np.load(os.path.join(golden_dir, "c1_filelist.npz"))
    ref_out = {}
    for name, cast in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        low = {k: torch.from_numpy(v).to(cast) for k, v in full_weights.items()}
        up = {k: v.to(torch.float32) for k, v in low.items()}
        a = make_net(); a.load_state_dict(up)                 # float32 host path
        b = make_net(); b.load_state_dict(low)                # typed host path
        c = make_net(); c.load_state_dict({k: v.cuda() for k, v in low.items()})   # typed device path
        oa, ob, oc = (golden_infer(n, g)[0] for n in (a, b, c))
        assert torch.equal(oa, ob) and torch.equal(oa, oc), name
        ref_out[name] = oa
    d = make_net(); d.load_state_dict({k: torch.from_numpy(v).double() for k, v in full_weights.items()})
    e = make_net(); e.load_state_dict(full_weights)
    assert torch.equal(golden_infer(d, g)[0], golden_infer(e, g)[0])
    assert rel_err(golden_infer(e, g)[0].cpu().numpy(), g["o"]) <= WAVE_TOL


@gpu
def test_second_load_replaces_folded_and_unfolded_forms(dims, full_weights, golden_dir):
    """a pre-folded '<x>.weight' load followed by a weight_g / weight_v load on the SAME model must use
    the second checkpoint (and the other way round)."""
    g = np.load(os.path.join(golden_dir, "c1_filelist.npz"))
    folded = {}
    for k, v in full_weights.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            gk = k[:-1] + "g"
            vv = v.astype(np.float64)
            nrm = np.sqrt((vv.reshape(vv.shape[0], -1) ** 2).sum(axis=1)).reshape(full_weights[gk].shape)
            folded[k[:-2]] = (vv * (full_weights[gk] / nrm)).astype(np.float32)
        else:
            folded[k] = v
    other = {k: (v * 0.5 if k.endswith("weight_g") else v) for k, v in full_weights.items()}
    a = make_net(); a.load_state_dict(folded, strict=False)
    o_folded = golden_infer(a, g)[0]
    assert rel_err(o_folded.cpu().numpy(), g["o"]) <= WAVE_TOL
    a.load_state_dict(other)                                   # same model, now weight_g / weight_v with other values
    b = make_net(); b.load_state_dict(other)
    assert torch.equal(golden_infer(a, g)[0], golden_infer(b, g)[0])
    assert not torch.equal(golden_infer(a, g)[0], o_folded)
    a.load_state_dict(folded, strict=False)                    # and back
    assert torch.equal(golden_infer(a, g)[0], o_folded)


@gpu
def test_adopted_arena_needs_a_commit_and_carries_what_rank0_packed(dims, full_weights):
    """has_voice_conversion of an adopting rank comes from the arena header, and the engine
    is not ready before the header has been checked."""
    from vispeech_amd._lib import VspError
    from vispeech_amd.engine import Engine
    from vispeech_amd.synth import synth_state_dict
    infer_only = synth_state_dict(dims, seed=1234, infer_only=True)
    for sd, want_vc in ((infer_only, False), (full_weights, True)):
        root = Engine(dims, "cuda:0")
        root.set_weights(sd)
        arena = root.finalize()
        assert root.has_voice_conversion == want_vc
        peer = Engine(dims, "cuda:0")
        mine = peer.adopt()
        mine.zero_()                                             # (fresh memory may hold a stale arena of this process)
        assert not peer.ready
        with pytest.raises(VspError):
            peer.commit_adopted()                                # nothing received yet: no header
        mine.copy_(arena)                                        # "the broadcast"
        torch.cuda.synchronize()
        peer.commit_adopted()
        assert peer.ready and peer.has_voice_conversion == want_vc
        z = torch.randn(1, dims.inter_channels, 20, device="cuda")
        gv = torch.randn(1, dims.gin_channels, device="cuda")
        assert torch.equal(peer.generator(z, gv), root.generator(z, gv))
    # a peer built for another configuration refuses the bytes
    import dataclasses
    other = Engine(dataclasses.replace(dims, n_speakers=dims.n_speakers + 1), "cuda:0")
    if other.arena_bytes() == root.arena_bytes():
        o_arena = other.adopt()
        o_arena.copy_(arena)
        with pytest.raises(VspError):
            other.commit_adopted()


@gpu
def test_first_streamed_chunk_leaves_long_before_the_whole_waveform(dims):
    """on the 60 s utterance of BASELINE config 5 the service's first PCM16 chunk (64 frames) is on the
    host well before the one-shot call -- what the reference's /tts does (inference_api.py:43-54) -- would return, and the
    streamed bytes are the one-shot waveform."""
    import time
    from vispeech_amd.service import SynthesisService
    from vispeech_amd.synth import synth_state_dict, workload
    net = make_net()
    net.load_state_dict(synth_state_dict(dims, seed=1234, infer_only=True))
    svc = SynthesisService(net, chunk_frames=64)
    batch = workload("C5")
    noise = torch.from_numpy(batch["noise"]).to(net.device)
    firsts, wholes = [], []
    for _ in range(3):                                   # (the first pass allocates workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = svc.stream(batch, 0, noise=noise)
        head = next(it)
        firsts.append(time.perf_counter() - t0)
        rest = b"".join(it)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pcm = svc.synthesize(batch, 0, noise=noise)
        wholes.append(time.perf_counter() - t1)
    assert len(head) == 2 * 64 * 512
    assert head + rest == pcm.tobytes()                  # byte-identical to the one-shot waveform
    assert min(firsts[1:]) < 0.6 * min(wholes[1:]), (firsts, wholes)


@gpu
def test_pooled_service_slots_deliver_the_single_service_audio(dims):
    """Two slots on two streams, driven from two threads at once: each request's PCM16 equals the one-context service's
    (same kernels, same inputs), streamed chunks included."""
    from vispeech_amd.pipeline import InFlightPool
    from vispeech_amd.service import PooledSynthesisService, SynthesisService
    from vispeech_amd.synth import synth_batch, synth_state_dict
    sd = synth_state_dict(dims, seed=1234, infer_only=True)
    pool = InFlightPool(make_net, lambda m: m.load_state_dict(sd), n=2)
    svc = PooledSynthesisService(pool, chunk_frames=32)
    one = SynthesisService(pool.nets[0], chunk_frames=32)
    batches = [synth_batch(2, seed=40 + i, mean_phonemes=12, std_phonemes=2, min_phonemes=8, max_phonemes=16, mean_frames=90,
                           jitter_frames=30) for i in range(4)]
    noises = [torch.from_numpy(b["noise"]).to(pool.nets[0].device) for b in batches]
    ref = [one.synthesize(b, 1, noise=n) for b, n in zip(batches, noises)]
    got = [None] * 4

    def work(k):
        r = None
        while r is None:                       # (a refused request is retried: only two slots)
            r = svc.synthesize(batches[k], 1, noise=noises[k])
        got[k] = r

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join(60)
    for k in range(4):
        np.testing.assert_array_equal(got[k], ref[k])
    streamed = b"".join(svc.stream(batches[0], 1, noise=noises[0]))
    assert streamed == ref[0].tobytes()
    assert not svc.busy
