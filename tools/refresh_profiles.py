#!/usr/bin/env python3
"""Copy the judged artifacts of gpurun_out/final/ (tools/final_profile.sh) into profiles/ under a tag.
usage: tools/refresh_profiles.py r02_final"""
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
for old in glob.glob(os.path.join(P, tag + "_*")):
    os.remove(old)
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(F, "trace", "t_kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(F, "trace_c5", "t_kernel_stats.csv")):
    shutil.copy(os.path.join(F, "trace_c5", "t_kernel_stats.csv"), os.path.join(P, f"{tag}_c5_kernel_stats.csv"))
shutil.copy(os.path.join(F, "generator_per_launch.txt"), os.path.join(P, f"{tag}_generator_per_launch.txt"))
shutil.copy(os.path.join(F, "pytest_gpu.txt"), os.path.join(P, f"{tag}_pytest_gpu.txt"))
for f in ("one_utterance_timeline.txt", "c5_timeline.txt", "operating_point_n8_timeline.txt", "operating_point_n8_generator.txt",
          "power.txt"):
    if os.path.exists(os.path.join(F, f)):
        shutil.copy(os.path.join(F, f), os.path.join(P, f"{tag}_{f}"))
shutil.copy(os.path.join(F, "traffic.json"), os.path.join(P, "traffic.json"))
if os.path.exists(os.path.join(F, "ttfa.json")):
    shutil.copy(os.path.join(F, "ttfa.json"), os.path.join(P, f"{tag}_time_to_first_audio.json"))
if os.path.exists(os.path.join(F, "two_rank_loop", "summary.txt")):
    shutil.copy(os.path.join(F, "two_rank_loop", "summary.txt"), os.path.join(P, f"{tag}_two_rank_loop.txt"))
# per kernel family: measured HBM bytes (PMC) beside the launches
tr = json.load(open(os.path.join(F, "traffic.json")))
with open(os.path.join(P, f"{tag}_hbm_traffic_per_kernel_family.txt"), "w") as fh:
    fh.write("HBM-side bytes per launch of the generator kernel families on the C3 batch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,\n"
             "KiB units, FETCH_SIZE x 2: MI355X_MICROARCH.md's gfx950 correction; tools/traffic_from_pmc.py).  Library " + tr.get("lib_sha256", "?")[:16] + "\n\n")
    fh.write(f"{'family':18s} {'kernel':12s} {'launches/step':>13s} {'fetch GB':>9s} {'write GB':>9s} {'total GB':>9s}\n")
    for k, v in (tr.get("families") or {}).items():
        fh.write(f"{k:18s} {v['kernel']:12s} {v['launches_per_step']:13.1f} {v['fetch_bytes_per_launch'] / 1e9:9.3f} {v['write_bytes_per_launch'] / 1e9:9.3f} {v['hbm_bytes_per_launch'] / 1e9:9.3f}\n")
    fh.write(f"{'all g16_* launches':18s} {'':12s} {tr['launches_per_step']:13.1f} {tr['fetch_bytes_per_launch'] / 1e9:9.3f} {tr['write_bytes_per_launch'] / 1e9:9.3f} {tr['hbm_bytes_per_launch'] / 1e9:9.3f}\n")
out = {}
for f, k in (("bench_c2", "C2"), ("bench_c5", "C5"), ("bench_controls_duration", "C3_duration_supplied_F0_energy_predicted"),
             ("bench_controls_none", "C3_all_predictors_on"), ("bench_f16mode", "C3_f16_reduced_precision_mode"),
             ("bench_traced", "C3_under_rocprofv3_trace"), ("bench_one_utterance", "one_utterance_latency"),
             ("bench_c4", "C4_256_utterances_on_one_gpu")):
    path = os.path.join(F, f + ".json")
    if not os.path.exists(path):
        continue
    x = json.loads(open(path).read().strip().splitlines()[-1])
    out[k] = {kk: x[kk] for kk in ("value", "unit", "ms_per_step", "rtf", "dtype")}
    out[k]["batches_in_flight"] = x["config"].get("batches_in_flight", 1)
    if x.get("single_batch"):
        out[k]["single_batch"] = {kk: x["single_batch"][kk] for kk in ("ms_per_step", "value", "rtf")}
    out[k]["workload"] = x["config"]["workload"]
    out[k]["padded_frames"] = x["config"]["padded_frames"]
    out[k]["valid_samples_per_step"] = x["config"]["valid_samples_per_step"]
    if "parity" in x:
        out[k]["parity"] = x["parity"]
    if "roofline" in x and "attention" in x["roofline"]:
        out[k]["attention"] = x["roofline"]["attention"]
        out[k]["generator_ms_per_step"] = x["roofline"]["kernel_ms_per_step"]
    print(k, round(x["ms_per_step"], 2), "ms", round(x["value"] / 1e6, 1), "M samples/s")
json.dump(out, open(os.path.join(P, f"{tag}_other_workloads.json"), "w"), indent=1)
# round 5: per-rank operating points (bench.py --shard-of N: rank 0's shard_range slice under the global padding) and the
# untrimmed second implementation, same box as the headline line
ops = {}
for N in (1, 2, 4, 8):
    path = os.path.join(F, f"bench_shard_of_{N}.json")
    if os.path.exists(path):
        x = json.loads(open(path).read().strip().splitlines()[-1]); r = x["roofline"]
        ops[f"N={N}"] = {"utterances_per_gpu": x["config"]["utterances_per_gpu"], "padded_frames": x["config"]["padded_frames"],
                         "ms_per_step": x["ms_per_step"], "samples_per_s_per_gpu": x["value"],
                         "batches_in_flight": x["config"].get("batches_in_flight", 1),
                         "single_batch_ms_per_step": (x.get("single_batch") or {}).get("ms_per_step"),
                         "single_batch_samples_per_s_per_gpu": (x.get("single_batch") or {}).get("value"),
                         "generator_ms": r["kernel_ms_per_step"], "generator_launches": r["launches"],
                         "attention_ms": r["attention"]["ms_per_step"], "frame_rate_convs_ms": r["frame_rate_convs"]["ms_per_step"],
                         "frame_rate_launches": r["frame_rate_convs"]["launches"],
                         "families_ms": (r.get("dominant_kernel") or {}).get("families_ms_per_step")}
if ops:
    base = ops.get("N=1", {}).get("samples_per_s_per_gpu")
    base1 = ops.get("N=1", {}).get("single_batch_samples_per_s_per_gpu")
    for k, v in ops.items():
        v["per_gpu_rate_vs_N1"] = v["samples_per_s_per_gpu"] / base if base else None
        v["single_batch_per_gpu_rate_vs_N1"] = (v["single_batch_samples_per_s_per_gpu"] / base1) if base1 and v.get("single_batch_samples_per_s_per_gpu") else None
        v["predicted_whole_job_samples_per_s"] = v["samples_per_s_per_gpu"] * int(k[2:])   # (before the gather: an upper bound)
    path = os.path.join(F, "bench_untrimmed.json")
    if os.path.exists(path):
        x = json.loads(open(path).read().strip().splitlines()[-1])
        ops["untrimmed_N=1 (VSP_TRIM_TAILS=0)"] = {"ms_per_step": x["ms_per_step"], "samples_per_s": x["value"],
                                                   "generator_ms": x["roofline"]["kernel_ms_per_step"]}
    json.dump(ops, open(os.path.join(P, f"{tag}_per_rank_operating_points.json"), "w"), indent=1)
    for k, v in ops.items():
        print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "families_ms"})
d = json.loads(open(os.path.join(F, "bench.json")).read().strip().splitlines()[-1])
r = d["roofline"]
print("C3", round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "M;",
      {k: r.get(k) for k in ("bound", "achieved", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_ms", "mfma_issue_frac", "hbm_measured_gbs")},
      "cpu", round(d["cpu_baseline"]["value"]), "parity", d.get("parity"))
