#!/bin/bash
# same-box per-launch times of g16_rw library variants (build/<dir>/libvispeech_hip.so; "product" = in-tree) at the C3
# size of the 32-channel stage.  usage: r04_rw_variants.sh <rounds> <dir|product> ...
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_rw_variants"; mkdir -p "$O"; N="$1"; shift
for i in $(seq 1 "$N"); do
  for D in "$@"; do
    if [ "$D" != "product" ]; then export VSP_LIB_PATH="$R/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
    for K in ${KS:-7 11}; do
      rm -rf "$O/t"
      rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o t -- python3 "$R/tools/pair_time.py" ${CH:-32} $K 1,3,5 > /dev/null 2>> "$O/err.txt"
      python3 - "$O/t" "$D" "$K" <<'PY' | tee -a "$O/table.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "g16_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = len(d) // 3
print(f"{sys.argv[2]:10s} K={sys.argv[3]:>2s}: " + "  ".join(f"{v:6.3f}" for v in d[-n:]) + "  ms   " + rows[-1]["Kernel_Name"][:44])
PY
    done
  done
done
