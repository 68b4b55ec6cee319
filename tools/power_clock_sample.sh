#!/bin/bash
# GPU clock / power while a bench workload runs (is the path power-limited?).
# usage (through gpurun): tools/power_clock_sample.sh <tag> "<bench.py arguments>" [ENV=val ..]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; TAG="$1"; ARGS="$2"; shift; shift
for kv in "$@"; do export "$kv"; done
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
( for i in $(seq 1 600); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.05; done ) > "$O/smi_samples.txt" &
SP=$!
# shellcheck disable=SC2086
python3 "$R/bench.py" $ARGS --no-cpu-baseline --profile-steps 0 > "$O/bench.json" 2> "$O/bench.err"
kill $SP 2>/dev/null; wait $SP 2>/dev/null
python3 - "$O/smi_samples.txt" "$O/bench.json" "$TAG" <<'PY'
import json, re, sys
rows = [l for l in open(sys.argv[1]) if l.strip()]
val = lambda pat, l: (lambda m: float(m.group(1)) if m else None)(re.search(pat, l))
s = [(val(r"sclk clock level: \S+ \((\d+)Mhz\)", l), val(r"Power \(W\): ([0-9.]+)", l), val(r"junction\) \(C\): ([0-9.]+)", l)) for l in rows]
busy = [x for x in s if x[1] and x[1] > 600]            # samples taken while the timed loop runs
med = lambda v: sorted(v)[len(v) // 2] if v else None
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[3]}: {d['ms_per_step']:.3f} ms/step over {d['steps']} steps | samples {len(s)} ({len(busy)} under load) | "
      f"sclk MHz median under load {med([x[0] for x in busy if x[0]])}, max seen {max([x[0] for x in s if x[0]] or [0])} | "
      f"power W median under load {med([x[1] for x in busy])}, max {max([x[1] for x in s if x[1]] or [0])} | junction C max {max([x[2] for x in s if x[2]] or [0])}")
PY
