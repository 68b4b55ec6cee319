#!/bin/bash
# End-of-round measurement set on the GPU box -> gpurun_out/final/ (tools/refresh_profiles.py copies what is judged
# into profiles/).  usage: tools/final_profile.sh
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
(cd "$R" && python -m pytest tests -q -m gpu 2>&1 | tail -3 > "$O/pytest_gpu.txt") || true
# (every rocprofv3 run below passes --in-flight 1: one stream, the launch sequence and step count the trace / PMC tools assume)
# PMC passes first (bench.py reads profiles/traffic.json written from them by refresh_profiles.py on the NEXT run;
# here the fresh numbers are merged into this run's bench line by refresh_profiles.py)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o p -- python3 "$R/bench.py" --in-flight 1 --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/bench.err" || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o p -- python3 "$R/bench.py" --in-flight 1 --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/bench.err" || true
python3 "$R/tools/traffic_from_pmc.py" 'g16_conv|g16_pair|g16_chain|g16_ups|g16_rw|g16_pp|g16_rc' "$O/pmc_fetch/p_counter_collection.csv" "$O/pmc_write/p_counter_collection.csv" f16s 64 489 > "$O/traffic.json" || true
cp "$O/traffic.json" "$R/profiles/traffic.json"   # the bench lines below report THIS build's measured traffic
python3 "$R/bench.py" --steps 20 --warmup 5 > "$O/bench.json" 2>> "$O/bench.err"
python3 "$R/bench.py" --workload C2 --steps 10 --cpu-runs 1 > "$O/bench_c2.json" 2>> "$O/bench.err"
python3 "$R/bench.py" --workload C5 --steps 10 > "$O/bench_c5.json" 2>> "$O/bench.err"
python3 "$R/bench.py" --controls duration --steps 10 --cpu-sample 4 > "$O/bench_controls_duration.json" 2>> "$O/bench.err"
python3 "$R/bench.py" --controls none --steps 5 --cpu-sample 2 > "$O/bench_controls_none.json" 2>> "$O/bench.err"
VSP_GENERATOR=f16 python3 "$R/bench.py" --cpu-sample 2 > "$O/bench_f16mode.json" 2>> "$O/bench.err"
python3 "$R/bench.py" --workload C2 --batch 1 --steps 50 --warmup 10 --no-cpu-baseline > "$O/bench_one_utterance.json" 2>> "$O/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python3 "$R/bench.py" --in-flight 1 --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline > "$O/bench_traced.json" 2>> "$O/bench.err" || true
python3 "$R/tools/trace_fused.py" "$O/trace/t_kernel_trace.csv" > "$O/generator_per_launch.txt" || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_c5" -o t -- python3 "$R/bench.py" --in-flight 1 --workload C5 --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline > "$O/bench_c5_traced.json" 2>> "$O/bench.err" || true
# one utterance and the 60 s utterance, launch by launch (tools/trace_timeline.py)
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_one" -o t -- python3 "$R/bench.py" --in-flight 1 --workload C2 --batch 1 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/bench.err" || true
python3 "$R/tools/trace_timeline.py" "$O/trace_one/t_kernel_trace.csv" > "$O/one_utterance_timeline.txt" || true
python3 "$R/tools/trace_timeline.py" "$O/trace_c5/t_kernel_trace.csv" > "$O/c5_timeline.txt" || true
# round 5: the per-rank operating points of the strong-scaling metric (rank 0's slice at N = 1 / 2 / 4 / 8, global padding)
# and the trimmed-tails switch, same box
for N in 1 2 4 8; do
  python3 "$R/bench.py" --shard-of $N --shard-rank 0 --steps 20 --warmup 4 --no-cpu-baseline > "$O/bench_shard_of_$N.json" 2>> "$O/bench.err" || true
done
VSP_TRIM_TAILS=0 python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_untrimmed.json" 2>> "$O/bench.err" || true
python3 "$R/bench.py" --workload C4 --steps 5 --warmup 2 --cpu-sample 2 --cpu-runs 1 > "$O/bench_c4.json" 2>> "$O/bench.err" || true
# round 6: launch-by-launch timelines at the small operating points, time to first audio, power / clock under load,
# per-family HBM bytes (profiles/traffic.json "families"), the two-rank loop with per-run verdicts
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_n8" -o t -- python3 "$R/bench.py" --in-flight 1 --shard-of 8 --shard-rank 0 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/bench.err" || true
python3 "$R/tools/trace_timeline.py" "$O/trace_n8/t_kernel_trace.csv" > "$O/operating_point_n8_timeline.txt" || true
python3 "$R/tools/trace_fused.py" "$O/trace_n8/t_kernel_trace.csv" 8 489 > "$O/operating_point_n8_generator.txt" 2>&1 || true
(cd "$R" && python3 tools/ttfa.py 64 > "$O/ttfa.json" 2>> "$O/bench.err") || true
(cd "$R" && tools/power_clock_sample.sh final/power_c3 "--steps 200 --warmup 5" > "$O/power.txt" 2>&1; tools/power_clock_sample.sh final/power_one "--workload C2 --batch 1 --steps 3000 --warmup 5" >> "$O/power.txt" 2>&1) || true
(cd "$R" && tools/loop_two_rank.sh 10 final/two_rank_loop > /dev/null 2>&1) || true
rm -f "$O"/trace*/t_kernel_trace.csv "$O"/pmc_*/p_agent_info.csv
cat "$O/pytest_gpu.txt"; cut -c1-300 "$O/bench.json"; tail -3 "$O/generator_per_launch.txt"; cat "$O/traffic.json"
