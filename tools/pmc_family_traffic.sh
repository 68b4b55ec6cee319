#!/bin/bash
# HBM bytes per launch of single generator kernel families (FETCH_SIZE / WRITE_SIZE passes + tools/traffic_from_pmc.py);
# usage (through gpurun): tools/pmc_family_traffic.sh
set -eu
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_nt"; rm -rf "$O"; mkdir -p "$O"
# (optionally: export VSP_LIB_PATH=$R/build/<dir>/libvispeech_hip.so to measure another build)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/f" -o p -- python3 "$R/bench.py" --in-flight 1 --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/err" || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/w" -o p -- python3 "$R/bench.py" --in-flight 1 --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2>> "$O/err" || true
for f in g16_pair g16_rw g16_rc; do python3 "$R/tools/traffic_from_pmc.py" "$f" "$O/f/p_counter_collection.csv" "$O/w/p_counter_collection.csv" f16s 64 489 | python3 -c "
import json,sys
d=json.load(sys.stdin); print('$f', d['launches_per_step'], 'fetch GB/launch', round(d['fetch_bytes_per_launch']/1e9,2), 'write', round(d['write_bytes_per_launch']/1e9,2))"; done
