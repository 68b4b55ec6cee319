#!/bin/bash
# step time, board power and clock of several library builds on ONE box, back to back.
# usage (through gpurun): tools/run_power_ab.sh <build dir | product> ...   (build/<dir>/libvispeech_hip.so)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/power_ab; mkdir -p $O
for D in "$@"; do
  if [ "$D" != "product" ]; then export VSP_LIB_PATH="$GRAFT_REPO_ROOT/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
  python bench.py --steps ${STEPS:-150} --warmup 3 --no-cpu-baseline --profile-steps 2 > $O/bench_$D.json 2> $O/bench_$D.err &
  BP=$!
  sleep 11
  for i in $(seq 1 12); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 1200; echo; sleep 0.4; done > $O/smi_$D.txt
  wait $BP
  python - <<PY
import json
pw, ck = [], []
for ln in open("$O/smi_$D.txt"):
    ln = ln.strip()
    if not ln.startswith("{"): continue
    try: d = json.loads(ln)
    except Exception: continue
    for k, v in d.items():
        for kk, vv in v.items():
            if "Package Power" in kk: pw.append(float(vv))
            if kk.startswith("sclk clock speed"): ck.append(float(vv.strip("()Mhz")))
b = json.loads(open("$O/bench_$D.json").read().strip().splitlines()[-1])
P = sum(pw) / max(len(pw), 1); t = b["ms_per_step"]
print(f"$D: {t:7.2f} ms/step  generator {b['roofline']['kernel_ms_per_step']:6.2f} ms  {P:6.0f} W  sclk {sum(ck)/max(len(ck),1):5.0f} MHz  -> {P*t/1000:6.1f} J/step")
PY
done
