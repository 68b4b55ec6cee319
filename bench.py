#!/usr/bin/env python3
"""Benchmark of the MI355X synthesis path: 44.1 kHz output samples/s (and RTF) of
SynthesizerTrn.infer on the BASELINE.json headline workload (C3: a 64-utterance mixed zh/ja
batch of ~5 s utterances, phoneme/duration/F0/energy/noise supplied), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--controls all|duration|none]

`--gpus N` with N > 1 launches itself: the parent process (which never touches a GPU) starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`
and relays rank 0's single JSON line; under torch.distributed.run (RANK / WORLD_SIZE in the environment) it is a
rank.  Workloads:
  C3 (default)  BASELINE.json's metric literally: ONE 64-utterance batch at 1 / 2 / 4 / 8 GPUs -- rank r synthesises
                its `shard_range` slice of the same batch (strong scaling; the N = 1 line is the single-GPU headline);
                `--weak` gives every rank its own 64 utterances instead (weak scaling, seeds 103 + rank);
  C4            ONE global batch of 256 utterances, rank r synthesises its `shard_range` slice (32 each at 8 GPUs;
                strong scaling);
all pad to the GLOBAL frame count (one int all-reduce MAX, SURVEY gotcha G6), broadcast the packed weights from
rank 0 over RCCL once before timing, and end every step with the gather of the waveforms on rank 0 -- the
exchange the north star names.  value = valid samples of all ranks / max-over-ranks time; the line also carries the
per-rank step times (min / max) and the gather's share of a step (a second timed pass without the gather).

A step = one full infer() of one batch with all inputs resident in HBM.  By default TWO batches are in flight per GPU
(--in-flight 2: two contexts on two HIP streams, step k on context k % 2 -- the frame-rate half of batch k + 1, ~120 short
launches that leave most of the chip idle, overlaps the generator of batch k); `value` is the throughput of the K timed
steps, `single_batch` the same K steps with one batch in flight (a step = one batch start to end: the latency figure).  The headline is timed with profiling OFF; the
per-kernel-class event timing behind `roofline` comes from a second, untimed pass of `--profile-steps` steps.
At N = 1 rank 0 then times the parity-pinned CPU oracle on a bounded sample of the same batch (`cpu_baseline`)
and checks the GPU waveform of those utterances against it (`parity`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# tools/micro/mfma_shape.hip on MI355X, pseudo-random operands, v_mfma_f32_16x16x32_f16 fed from LDS, barrier per 64-deep step
MEASURED_F16_MFMA_CEILING_TFLOPS = 1844.0
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
ALG_BYTES_PER_SAMPLE = 13045     # SURVEY.md 8(d), fp32 end-to-end layer-boundary traffic
ALG_FLOPS_PER_SAMPLE = 1.641e6

WL_TEXT = {"C1": "zh utterance (~5 s)", "C2": "zh utterances (~5 s each)", "C3": "mixed zh/ja utterances (~5 s each)",
           "C4": "mixed zh/ja utterances (~5 s each), one global batch sharded over the ranks",
           "C5": "long-form utterance (60 s, 5168 frames)"}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="C3")
    p.add_argument("--batch", type=int, default=None,
                   help="override the batch: utterances per GPU (C3 and the other per-rank workloads) or of the GLOBAL batch (C4)")
    p.add_argument("--controls", default="all", choices=["all", "duration", "none"],
                   help="which control tensors are supplied: all (headline), duration only (F0 / energy predicted), "
                        "none (every predictor of reference models.py:681-708 runs)")
    p.add_argument("--profile-steps", type=int, default=2, help="untimed steps with per-launch events (roofline)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=16, help="utterances of the batch timed on the CPU oracle")
    p.add_argument("--cpu-runs", type=int, default=3, help="timed CPU-oracle runs (the median is reported) after one warm-up run")
    p.add_argument("--weak", action="store_true",
                   help="C3 and the other per-rank workloads: every rank synthesises its OWN batch (weak scaling) instead of "
                        "its shard_range slice of ONE batch (strong scaling, the default: BASELINE.json's metric is one "
                        "64-utterance batch at 1 / 2 / 4 / 8 GPUs)")
    p.add_argument("--shard-of", type=int, default=0,
                   help="one GPU, no process group: synthesise ONLY rank --shard-rank's shard_range slice of the batch under the "
                        "GLOBAL frame padding -- the per-GPU operating point a --gpus N run lands on (profiles/r05_per_rank_*)")
    p.add_argument("--shard-rank", type=int, default=0)
    p.add_argument("--in-flight", type=int, default=2,
                   help="batches in flight per GPU: N contexts on N HIP streams, step k on context k %% N, so that the frame-rate "
                        "half of batch k + 1 (~120 short launches that leave most of the chip idle) overlaps the generator of "
                        "batch k.  Every step is still one full infer() of one batch; `value` is the throughput of the K timed "
                        "steps.  The line also carries `single_batch`: the same K steps with ONE batch in flight (--in-flight 1: "
                        "a step = one batch start to end, the latency figure and rounds 1-5's headline)")
    p.add_argument("--dump-wave", default=None,
                   help="test hook: rank 0 writes the waveform batch of one more (untimed) step -- gathered over the ranks when "
                        "there is a process group -- to this .npy file")
    p.add_argument("--dist", action="store_true",
                   help="initialise torch.distributed (RCCL) even at --gpus 1: the weight broadcast, the frame-count "
                        "all-reduce and the waveform gather run through the collectives with one rank")
    return p.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILD processes (this
    process has not touched the GPU and never will) and relay their output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env)
    return p.returncode


def usable_cores() -> int:
    """Threads actually usable by this process (cgroup / affinity), capped: torch's CPU convolutions stop scaling
    (and collapse under oversubscription) far below the 256 hardware threads of the host."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota of the box ("max" or "<quota> <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, min(avail, 32))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def lib_sha256() -> str:
    import hashlib
    from vispeech_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def cpu_baseline_and_parity(sd, dims, batch, n_utt, controls, tf_global, noise_global, gpu_out, runs=3):
    """The parity-pinned CPU restatement (oracle/, kind 'port') on the first n_utt utterances of the same batch,
    padded exactly like the GPU run (global T_p and T_f, gotchas G5 / G6) so that its waveform is THE reference for
    the GPU output of those utterances.  Returns (cpu_baseline, parity)."""
    import numpy as np
    import torch
    from oracle.vispeech_oracle import Oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    orc = Oracle(sd, dims)
    sl = slice(0, n_utt)
    kw = dict(noise_scale=0.667)
    if controls in ("all", "duration"):
        kw["duration_control"] = batch["duration"][sl]
    if controls == "all":
        kw["pitch_control"] = batch["f0"][sl]
        kw["energy_control"] = batch["energy"][sl]
    # warm-up run (1 warm-up + median of >= 3): the same call on the first two utterances --
    # thread pool, mkldnn primitives and the allocator see the real shapes
    w = slice(0, min(2, n_utt))
    orc.infer(batch["phonemes"][w], batch["lengths"][w], batch["sid"][w], noise=noise_global[w], t_f=tf_global,
              **{k: (v[w] if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    times, ref = [], None
    for _ in range(max(1, runs)):
        t0 = time.perf_counter()
        ref = orc.infer(batch["phonemes"][sl], batch["lengths"][sl], batch["sid"][sl], noise=noise_global[sl],
                        t_f=tf_global, **kw)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    frames = np.asarray(gpu_out["frames"][sl], dtype=np.int64)
    samples = 512 * int(frames.sum())
    base = {"value": samples / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "host_threads": os.cpu_count(), "torch_threads": cores,
            "runs_s": [round(t, 2) for t in times], "timing": f"1 warm-up run + median of {len(times)} timed runs",
            "sample": f"first {n_utt} utterances of the same batch, padded to the batch's {tf_global} frames "
                      f"({samples} valid samples, median {dt:.1f} s, oracle/vispeech_oracle.py on torch CPU fp32, {cores} threads)",
            "rtf": dt / (samples / 44100.0)}
    # parity of the very utterances the timed run produced (outside the timed region)
    ro = ref["o"].numpy()
    go = gpu_out["o"][sl].cpu().numpy()
    par = {"utterances": int(n_utt), "padded_frames": int(tf_global), "tolerance": 1e-4}
    rd = np.asarray(ref["duration"]).reshape(n_utt, -1)
    gd = gpu_out["duration"][sl].cpu().numpy().reshape(n_utt, -1)
    par["duration_mismatches"] = int((rd != gd).sum())
    if go.shape == ro.shape and par["duration_mismatches"] == 0:
        par["max_rel_err"] = float(np.abs(go - ro).max() / max(np.abs(ro).max(), 1e-30))
        gz, rz = gpu_out["z"][sl].cpu().numpy(), ref["z"].numpy()
        par["z_max_rel_err"] = float(np.abs(gz - rz).max() / max(np.abs(rz).max(), 1e-30))
        par["ok"] = bool(par["max_rel_err"] <= par["tolerance"])
    else:
        par["max_rel_err"] = None
        par["ok"] = False
        par["note"] = f"shapes {go.shape} vs {ro.shape}: predicted durations differ, waveforms not comparable"
    return base, par


def c5_prefix_parity(sd, dims, batch, last, tf_global, frames=600):
    """Long-form workload: the CPU oracle's vocoder on the first `frames` frames of the GPU's own z (the generator is
    local: samples further than its receptive field -- 14 frames -- from the cut are exact) against the same samples
    of the GPU waveform; the frame-rate stages are covered by z itself in tests/test_baseline_configs.py."""
    import numpy as np
    import torch
    from oracle.vispeech_oracle import Oracle, generator
    orc = Oracle(sd, dims)
    n = min(frames, tf_global)
    z = last["z"][:1, :, :n].cpu()
    g = orc.w["emb_g.weight"][torch.as_tensor(batch["sid"][:1])].unsqueeze(-1)
    torch.set_num_threads(usable_cores())
    t0 = time.perf_counter()
    ref = generator(orc.w, z, g, dims).numpy()
    dt = time.perf_counter() - t0
    keep = 512 * (n - (16 if n < tf_global else 0))
    got = last["o"][:1, :, :keep].cpu().numpy()
    err = float(np.abs(got - ref[:, :, :keep]).max() / max(np.abs(ref).max(), 1e-30))
    return {"what": f"oracle vocoder on frames [0, {n}) of the GPU's z vs the GPU waveform, first {keep} samples",
            "max_rel_err": err, "tolerance": 1e-4, "ok": bool(err <= 1e-4), "oracle_s": round(dt, 1)}


def main():
    args = parse()
    if "RANK" not in os.environ and (args.gpus > 1 or args.dist):
        sys.exit(self_launch(args))
    # a rank that hangs must END, loudly and non-zero (torch.distributed.run then takes the other ranks and the parent
    # down with it): after VSP_BENCH_DUMP_AFTER seconds (default 900) every thread's Python stack goes to stderr and the
    # process exits -- a hung collective never outlives the run, and the dump says where it hung
    import faulthandler
    dump_after = int(os.environ.get("VSP_BENCH_DUMP_AFTER", "900"))

    def arm(extra: float = 0.0):
        """(Re)start the hang watchdog for the next PHASE: it is a hang detector, not a deadline for the whole run -- every
        phase gets its own window (a healthy long run, --steps 10000 or a slow CPU baseline, must not be killed)."""
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(dump_after + int(extra), exit=True)

    arm()
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from vispeech_amd import _lib, config as vcfg
    from vispeech_amd.models import SynthesizerTrn
    from vispeech_amd.schema import ModelDims
    from vispeech_amd.sharding import BatchGatherer, broadcast_weights, global_max, shard_counts, shard_range
    from vispeech_amd.synth import WORKLOADS, synth_batch, synth_state_dict

    # one process per GPU; VSP_BENCH_BACKEND=gloo (test hook) lets several ranks share one GPU to exercise the
    # multi-process path on a single-GPU box -- RCCL itself refuses two ranks on one device
    backend = os.environ.get("VSP_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.dist          # --dist: the collectives run (through RCCL) even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that does not complete raises after this timeout instead of hanging the run
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("VSP_BENCH_PG_TIMEOUT", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    n_ranks_seen = dist.get_world_size() if use_dist else 1

    dims = ModelDims()
    a, kw = vcfg.synthesizer_args(vcfg.default_hparams())
    net = SynthesizerTrn(*a, device=dev, **kw).eval()
    sd = synth_state_dict(dims, seed=1234, infer_only=True) if rank == 0 or world == 1 else None
    if not use_dist:
        net.load_state_dict(sd)
    else:
        # RCCL weight broadcast: rank 0 packs, everyone else adopts the broadcast arena and checks its header
        broadcast_weights(net._engine, sd, src=0)
        torch.cuda.synchronize()
    eng = net._engine
    # --in-flight N: N - 1 more contexts (their own packed weights and workspaces), each on its own stream
    if args.in_flight < 1:
        raise SystemExit("--in-flight N >= 1")
    from vispeech_amd.pipeline import InFlightPool

    def load_more(m):
        if not use_dist:
            m.load_state_dict(sd)
        else:
            broadcast_weights(m._engine, sd, src=0)      # (every rank takes part: same number of contexts everywhere)
            torch.cuda.synchronize()

    pool_all = InFlightPool(lambda: SynthesizerTrn(*a, device=dev, **kw).eval(), load_more, n=args.in_flight, first=net)
    pool = [pool_all]                                    # (the pool the steps run on: all contexts, or context 0 alone)

    # ---- the batch.  C4: one global batch, this rank's slice; otherwise one batch per rank
    wl = dict(WORKLOADS[args.workload])
    sharded_global = args.workload == "C4" or (not args.weak and args.workload != "C5")   # (C5 is ONE utterance: replicas only)
    if args.batch:
        wl["batch"] = args.batch
    if not sharded_global:
        wl["seed"] = wl["seed"] + rank
    batch = synth_batch(**wl)
    B_all = int(batch["phonemes"].shape[0])
    lo, hi = shard_range(B_all, rank, world) if sharded_global else (0, B_all)
    if args.shard_of:
        if world != 1 or args.workload == "C5" or args.controls == "none" or args.weak:
            raise SystemExit("--shard-of: one GPU, a sharded workload, durations supplied")
        sharded_global = True
        lo, hi = shard_range(B_all, args.shard_rank, args.shard_of)
    sl = slice(lo, hi)
    B = hi - lo
    t = lambda x: torch.from_numpy(np.asarray(x)).to(dev)
    ph, ln, sid = t(batch["phonemes"][sl]), t(batch["lengths"][sl]), t(batch["sid"][sl])
    ctl = {}
    if args.controls in ("all", "duration"):
        ctl["duration_control"] = t(batch["duration"][sl])
    if args.controls == "all":
        ctl["pitch_control"] = t(batch["f0"][sl])
        ctl["energy_control"] = t(batch["energy"][sl])
    # frame counts: known from the supplied durations, else from one untimed encode (the predictor decides)
    if "duration_control" in ctl:
        frames_local = batch["frame_lengths"][sl].astype(np.int64)
    else:
        enc = eng.encode(ph, ln, sid)
        frames_local = np.asarray(eng.frame_lengths_host(enc["frame_lengths"])[0], dtype=np.int64)
    tf_local = int(frames_local.max()) if B else 0
    tf_global = global_max(tf_local, dev)
    if args.shard_of:
        tf_global = int(batch["frame_lengths"].max())          # what the all-reduce MAX of the N ranks would return
    valid_samples = 512 * int(frames_local.sum())
    # frames the generator computes on this rank (trimmed tails: an utterance runs to length + back + 1 + fwd frames, the
    # rest of its padded waveform is filled from the steady state) -- the work the roofline figures are charged for
    back_f, fwd_f = eng.generator_frame_dependence()
    trim_on = os.environ.get("VSP_TRIM_TAILS", "1") != "0" and os.environ.get("VSP_GENERATOR", "f16s") != "f32"
    frames_computed = int(np.minimum(frames_local + back_f + 1 + fwd_f, tf_global).sum()) if trim_on else int(B * tf_global)
    r = np.random.Generator(np.random.PCG64(wl["seed"] * 7919 + 13))
    noise_np = np.zeros((B_all if sharded_global else B, dims.inter_channels, tf_global), dtype=np.float32)
    if "duration_control" in ctl:
        noise_np[:, :, :batch["noise"].shape[2]] = batch["noise"]
    else:
        noise_np[:] = r.standard_normal(noise_np.shape, dtype=np.float32)
    noise = t(noise_np[sl] if sharded_global else noise_np)

    last = {}
    # final waveform gather on rank 0 (RCCL over xGMI): shard sizes are known from shard_range, the receive buffers
    # are persistent, and the collective runs on a side stream so that step k + 1 overlaps the gather of step k
    gatherer = None
    if use_dist:
        counts = shard_counts(B_all, world) if sharded_global else [B] * world
        gatherer = BatchGatherer(counts, (1, 512 * tf_global), torch.float32, dev, dst=0)

    def step():
        # (the gather is started INSIDE the step's stream: it orders itself behind that stream's work by an event)
        g = gatherer
        (o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy), _ = pool[0].infer(
            ph, ln, sid=sid, noise_scale=0.667, noise=noise, t_f=tf_global,
            after=(lambda res: g.start(res[0])) if g is not None else None, **ctl)
        last.update(o=o, z=z, duration=duration)
        return o

    def drain():
        if gatherer is not None:
            gatherer.wait()                 # the last gather belongs to the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    arm()
    t_w0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    eng.profile(False)
    drain()
    # the timed phase's window scales with its length: the warm-up's own per-step time x steps, with a wide margin
    per_step_guess = (time.perf_counter() - t_w0) / max(args.warmup, 1)
    arm(extra=4.0 * per_step_guess * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    dt = time.perf_counter() - t0
    arm(extra=4.0 * per_step_guess * args.steps)
    dt_rank = dt
    # ONE batch in flight (a step = one batch start to end: the latency figure, and the headline of rounds 1-5): the same K
    # steps once more on context 0 alone (untimed for `value`)
    dt_single = None
    if len(pool_all) > 1:
        pool[0] = pool_all.restrict(1)
        for _ in range(2):
            step()
        drain()
        arm(extra=4.0 * per_step_guess * args.steps)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        dt_single = time.perf_counter() - t1
        pool[0] = pool_all
        arm(extra=4.0 * per_step_guess * args.steps)
    # the gather's share of a step: the same K steps once more WITHOUT the exchange (untimed for `value`)
    dt_nogather = None
    if gatherer is not None:
        keep, gatherer = gatherer, None
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        dt_nogather = time.perf_counter() - t1
        gatherer = keep

    # ---- second, untimed pass: HIP events around every launch of the profiled classes (rank 0's numbers are reported)
    prof = {}
    if args.profile_steps > 0:
        arm()
        pool[0] = pool_all.restrict(1)                 # (the per-launch event pass runs one batch at a time on context 0)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(args.profile_steps):
            step()
        drain()
        fams = eng.profile_read_families(_lib.PROF_GENERATOR)
        for name, cls in (("generator", _lib.PROF_GENERATOR), ("attention", _lib.PROF_ATTENTION), ("frame", _lib.PROF_FRAME)):
            n, ms, fl, by, bx, bm = eng.profile_read(reset=True, cls=cls)
            prof[name] = dict(launches=n // args.profile_steps, ms=ms / args.profile_steps, flops=fl / args.profile_steps,
                              bytes=by / args.profile_steps, bytes_ext=bx / args.profile_steps, moved=bm / args.profile_steps)
        if fams:
            # the kernel family with the largest share of the step (HIP events around each of its launches): what a reader
            # recomputes from profiles/*kernel_stats.csv (name, launches per step, average duration)
            KN = {("pair", 128): "g16_pp (ResBlock conv pairs, 128 channels)", ("pair", 64): "g16_pair<2,2,3,8,2> (ResBlock conv pairs, 64 channels)",
                  ("pair", 32): "g16_rw (ResBlock conv pairs, 32 channels, weights in registers)",
                  ("chain", 32): "g16_rc (whole k3 ResBlock, 32 channels)", ("conv", 256): "g16_conv (256-channel ResBlock convolutions; g16_convp on uniform batches)",
                  ("conv", 128): "g16_conv (128-channel k11 ResBlock convolutions)"}
            f = fams[0]
            k = args.profile_steps
            avg = f["ms"] / max(f["launches"], 1)
            prof["dominant"] = {
                "name": KN.get((f["kind"], f["channels"]), f"{f['kind']} C={f['channels']}"), "family": f"{f['kind']}{f['channels']}",
                "launches": f["launches"] // k, "avg_launch_ms": avg, "ms_per_step": f["ms"] / k,
                "alg_flops_per_launch": f["flops"] / max(f["launches"], 1), "alg_bytes_per_launch": f["bytes"] / max(f["launches"], 1),
                "alg_tflops": f["flops"] / max(f["ms"], 1e-9) / 1e9, "alg_gbs": f["bytes"] / max(f["ms"], 1e-9) / 1e6,
                "alg_frac_mfma": f["flops"] / max(f["ms"], 1e-9) / 1e9 / PEAK_F16_MFMA_TFLOPS,
                "alg_frac_hbm": f["bytes"] / max(f["ms"], 1e-9) / 1e6 / PEAK_HBM_GBS,
                # what the launch must move through HBM AS FUSED (input, output, residual, previous sum once each): the
                # byte figure that bounds a fused launch -- SURVEY 8d's layer-boundary model above charges a fused pair
                # the four passes of the two convolutions it replaces and is the contract's algorithmic figure only
                "moved_bytes_per_launch": f["moved"] / max(f["launches"], 1),
                "moved_gbs": f["moved"] / max(f["ms"], 1e-9) / 1e6,
                "moved_frac_hbm": f["moved"] / max(f["ms"], 1e-9) / 1e6 / PEAK_HBM_GBS,
                "mfma_issue_frac": 3.0 * f["flops"] / max(f["ms"], 1e-9) / 1e9 / PEAK_F16_MFMA_TFLOPS,
                "families_ms_per_step": {f"{x['kind']}{x['channels']}": round(x["ms"] / k, 3) for x in fams},
                "families": {f"{x['kind']}{x['channels']}": {
                    "launches": x["launches"] // k, "avg_launch_ms": round(x["ms"] / max(x["launches"], 1), 4),
                    "alg_gbs": round(x["bytes"] / max(x["ms"], 1e-9) / 1e6, 1),
                    "moved_gbs": round(x["moved"] / max(x["ms"], 1e-9) / 1e6, 1),
                    "alg_tflops": round(x["flops"] / max(x["ms"], 1e-9) / 1e9, 1)} for x in fams}}
        eng.profile(False)
        if use_dist:
            dist.barrier()

    if args.dump_wave:
        # one more untimed step on every rank; rank 0 keeps what the exchange delivered (the test compares an N-rank RCCL
        # run with the one-GPU run of the same batch)
        arm()
        step()
        shards = gatherer.wait() if gatherer is not None else [last["o"]]
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if rank == 0:
            np.save(args.dump_wave, torch.cat([x.reshape(x.shape[0], -1) for x in shards], dim=0).cpu().numpy())

    tt = torch.tensor([dt, float(valid_samples), dt_nogather or dt, dt_single or dt], dtype=torch.float64, device=dev)
    rank_ms = [dt_rank / args.steps * 1e3]
    if use_dist:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        mn = tt.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dt, total_valid = float(mx[0].item()), float(sm[1].item())
        rank_ms = [float(mn[0].item()) / args.steps * 1e3, float(mx[0].item()) / args.steps * 1e3]
        dt_nogather = float(mx[2].item())
        if dt_single is not None:
            dt_single = float(mx[3].item())
    else:
        total_valid = float(valid_samples)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_valid * args.steps / dt
        audio_s = total_valid / 44100.0
        gen_mode = os.environ.get("VSP_GENERATOR", "f16s")
        # (at one GPU the whole batch is that GPU's: the single-GPU headline line reads the same in both modes)
        split_text = " per GPU" if (not sharded_global or (world == 1 and args.workload != "C4")) else \
            (f", ONE batch split over {world} GPUs (shard_range)" if args.workload != "C4" else "")
        out = {
            "metric": "44.1kHz samples/sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if sharded_global else "weak", "vs_baseline": None,
            "dtype": "f32 (split-f16 MFMA, 3-term)", "data": "synthetic",
            "rtf": (dt / args.steps) / audio_s,
            "config": {"workload": f"{args.workload}: {B_all} {WL_TEXT.get(args.workload, 'utterances')}"
                                   f"{split_text}, 44.1 kHz, hop 512, "
                                   f"{ {'all': 'phoneme/duration/F0/energy/noise supplied', 'duration': 'phoneme/duration/noise supplied, F0 and energy PREDICTED', 'none': 'phonemes + noise supplied, duration / F0 / energy PREDICTED'}[args.controls] }"
                                   ", random-init (synthetic) weights of configs/config.json"
                                   + (f"; {args.in_flight} batches in flight per GPU (single_batch = one)" if args.in_flight > 1 else ""),
                       "utterances_per_gpu": B, "global_batch": B_all * (1 if sharded_global else world),
                       "padded_frames": tf_global, "valid_samples_per_step": int(total_valid),
                       "frames_padded": int(B * tf_global), "frames_computed": frames_computed,
                       "parallelism": f"shard{world}", "generator": gen_mode, "controls": args.controls,
                       "batches_in_flight": args.in_flight},
            "single_batch": ({"what": "the same K steps with ONE batch in flight (context 0 alone): a step = one batch start to end",
                              "ms_per_step": dt_single / args.steps * 1e3, "value": total_valid * args.steps / dt_single,
                              "rtf": (dt_single / args.steps) / audio_s} if dt_single is not None else None),
            "emulated_rank": ({"rank": args.shard_rank, "of": args.shard_of, "utterances": [lo, hi]} if args.shard_of else None),
            "n_ranks_seen": n_ranks_seen,
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            "gather": ({"bytes_into_rank0_per_step": int(4 * 512 * tf_global * (B_all * (1 if sharded_global else world) - B)),
                        "ms_per_step_without_gather": dt_nogather / args.steps * 1e3,
                        "share_of_step": max(0.0, 1.0 - dt_nogather / dt)} if use_dist else None),
            "collectives": (f"torch.distributed backend {dist.get_backend()}: weight-arena broadcast, frame-count all-reduce MAX, "
                            "waveform gather on rank 0 (side stream, persistent buffers)") if use_dist else "none (one process, no process group)",
        }
        if gen_mode == "f32":
            out["dtype"] = "f32"
        elif gen_mode == "f16":      # opt-in reduced precision: NOT the headline configuration
            out["dtype"] = "f16 operands, f32 accumulate (VSP_GENERATOR=f16: reduced precision, fails the fp32 parity gate)"
        if prof:
            g = prof["generator"]
            tfl = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0           # algorithmic TFLOP/s
            gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0            # SURVEY 8d bytes / time
            mvs = g["moved"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0            # bytes moved as fused / time
            mf = {"f32": 1.0, "f16": 1.0}.get(gen_mode, 3.0)                               # MFMA FLOPs issued per algorithmic FLOP
            peak = PEAK_F32_MFMA_TFLOPS if gen_mode == "f32" else PEAK_F16_MFMA_TFLOPS
            f_hbm, f_issue, f_alg, f_moved = gbs / PEAK_HBM_GBS, mf * tfl / peak, tfl / peak, mvs / PEAK_HBM_GBS
            # Which roof.  The work charged follows the work DONE: a trimmed ragged batch is charged
            # the frames each launch computes (config.frames_computed of frames_padded).  `bound` is the resource the
            # launches actually saturate first: the ISSUED matrix rate (three f16 MFMAs per fp32-accurate product in the
            # default mode) against the HBM rate of the bytes the launches MOVE as fused (each operand once; PMC-measured
            # traffic where profiles/traffic.json has it for this build).  SURVEY 8d's layer-boundary byte model stays in
            # the line as the contract's algorithmic figure (`hbm_frac`, `alg_gbs`), but it is not a bound of fused
            # launches -- single fused launches exceed the HBM peak under it -- so it never decides `bound`.
            # `frac` = ALGORITHMIC work of the bounding resource / its peak, never the issued rate.
            mfma_bound = f_issue >= f_moved
            roof = ({"bound": "mfma", "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": f_alg}
                    if mfma_bound else
                    {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_hbm})
            roof.update({
                "traffic": None,
                "bound_reason": (f"issued matrix work {f_issue:.3f} of the dense peak ({mf:.0f} MFMA per algorithmic product) against "
                                 f"{f_moved:.3f} of the HBM peak for the bytes the fused launches move; the layer-boundary byte model "
                                 f"({f_hbm:.3f}) is the contract's algorithmic figure, not a bound of fused launches"),
                "kernel": "generator convolutions (g16_conv / g16_convp: one launch per convolution; g16_pp / g16_pair / g16_rw: fused ResBlock conv pairs of the 128- / 64- / 32-channel stages; g16_rc: whole k3 ResBlocks of the 32-channel stage; g16_ups: the streaming up-convs; conv_post), rank 0",
                "launches": g["launches"], "kernel_ms_per_step": g["ms"], "avg_launch_ms": g["ms"] / max(g["launches"], 1),
                "alg_tflops": tfl, "alg_frac": tfl / peak, "alg_flops_per_launch": g["flops"] / max(g["launches"], 1),
                "alg_gbs": gbs, "alg_bytes_per_launch": g["bytes"] / max(g["launches"], 1),
                "alg_gbs_with_residual_reads": g["bytes_ext"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0,
                "moved_gbs": mvs, "moved_bytes_per_launch": g["moved"] / max(g["launches"], 1), "moved_frac_hbm": f_moved,
                "hbm_frac": f_hbm, "mfma_issue_frac": f_issue, "mfma_issue_tflops": mf * tfl,
                "work_charged": "frames computed (trimmed tails): config.frames_computed / config.frames_padded of the padded tensor",
                "dominant_kernel": prof.get("dominant"),
                "mfma_frac_of_measured_random_data_ceiling": mf * tfl / MEASURED_F16_MFMA_CEILING_TFLOPS if gen_mode != "f32" else None,
                "hbm_frac_whole_path": value / world * ALG_BYTES_PER_SAMPLE / (PEAK_HBM_GBS * 1e9),
                "note": "alg_gbs / hbm_frac count SURVEY 8d's layer-boundary bytes (input once + output once per convolution; "
                        "a fused pair = the two convolutions it replaces = 4 passes) and moved_* the bytes a launch moves as "
                        "fused, both for the frames the launches compute, measured with HIP events around every launch in a "
                        "separate untimed pass; hbm_frac_whole_path = valid samples/s x 13,045 B (SURVEY 8d's own formula); "
                        "fp32 activations in HBM, f16 MFMA on split operands (3 MFMAs per product) in the default mode",
            })
            at, fr = prof["attention"], prof["frame"]
            if at["ms"] > 0:
                atf = at["flops"] / (at["ms"] * 1e-3) / 1e12
                att_f32 = os.environ.get("VSP_ATT") == "f32"
                roof["attention"] = {
                    "kernel": ("attn_relpos_f32 (two passes, v_mfma_f32_32x32x2_f32)" if att_f32 else
                               "attn_pack_f16s + attn_relpos_f16s (one pass, v_mfma_f32_16x16x32_f16 on split operands: 3 MFMAs per product)")
                              + ", reference attentions.py:148-179",
                    "launches": at["launches"], "ms_per_step": at["ms"], "alg_tflops": atf,
                    "mfma_issue_tflops": atf if att_f32 else 3.0 * atf,
                    "mfma_peak_tflops": PEAK_F32_MFMA_TFLOPS if att_f32 else PEAK_F16_MFMA_TFLOPS,
                    "mfma_utilisation": atf / PEAK_F32_MFMA_TFLOPS if att_f32 else 3.0 * atf / PEAK_F16_MFMA_TFLOPS,
                    "flops_model": "4 H T^2 + 4 H T (2 window + 1) per utterance and layer"}
            if fr["ms"] > 0:
                roof["frame_rate_convs"] = {"launches": fr["launches"], "ms_per_step": fr["ms"],
                                            "alg_tflops": fr["flops"] / (fr["ms"] * 1e-3) / 1e12}
            # measured HBM traffic of the generator launches from an earlier rocprofv3 --pmc pass of THIS build and
            # workload (profiles/traffic.json: keyed on generator mode, utterances and padded frames)
            # and on the SHA-256 of the library the passes ran: a rebuilt library drops the stale figure instead of carrying it)
            traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(traffic_file) and not any(os.environ.get(k) for k in ("VSP_FUSE_PAIRS", "VSP_CHAIN")):
                try:
                    tr = json.load(open(traffic_file))
                    if tr.get("generator") == gen_mode and tr.get("utterances") == B and tr.get("padded_frames", tf_global) == tf_global \
                            and tr.get("lib_sha256") == lib_sha256():
                        roof["traffic"] = tr["hbm_bytes_per_launch"]
                        roof["traffic_source"] = "previous PMC pass: " + str(tr.get("source"))
                        roof["traffic_launches"] = tr.get("launches_per_step")   # (the g16_* launches: conv_pre runs on the frame-rate kernel, outside the PMC kernel filter)
                        roof["hbm_measured_gbs"] = tr["hbm_bytes_per_launch"] * tr.get("launches_per_step", g["launches"]) / (g["ms"] * 1e-3) / 1e9
                        # per-family PMC bytes (tools/pmc_family_traffic.sh): the dominant kernel's MEASURED traffic beside its models
                        dk = roof.get("dominant_kernel")
                        fam_tr = (tr.get("families") or {}).get(dk["family"]) if dk else None
                        if fam_tr:
                            dk["measured_hbm_bytes_per_launch"] = fam_tr["hbm_bytes_per_launch"]
                            dk["measured_hbm_gbs"] = fam_tr["hbm_bytes_per_launch"] / (dk["avg_launch_ms"] * 1e-3) / 1e9
                            dk["measured_frac_hbm"] = dk["measured_hbm_gbs"] / PEAK_HBM_GBS
                except Exception:
                    pass
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.workload == "C5":
            out["parity"] = c5_prefix_parity(sd, dims, batch, last, tf_global)
        elif world == 1 and not args.no_cpu_baseline:
            frames_all = frames_local
            gpu_out = dict(o=last["o"], z=last["z"], duration=last["duration"] if torch.is_tensor(last["duration"]) else ctl["duration_control"],
                           frames=frames_all)
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(
                sd, dims, {k: v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (B_all,) else v for k, v in batch.items()},
                min(args.cpu_sample, B), args.controls, tf_global, noise_np[sl] if sharded_global else noise_np, gpu_out,
                runs=args.cpu_runs)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
