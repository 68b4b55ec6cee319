#!/bin/bash
# timing-only ablations of g16_pp (VSP_PP_DIAG bits, gen16_pp.hip) at the C3 size of the 128-channel stage; usage: pp_diag.sh <diag..>
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pp_diag"; mkdir -p "$O"
for D in "$@"; do
  for K in 3 7; do
    rm -rf "$O/t"; export VSP_PP_DIAG="$D"
    rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o t -- python3 "$R/tools/pair_time.py" 128 $K 1,3,5 64 31296 > /dev/null 2>> "$O/err.txt"
    python3 - "$O/t" "$D" "$K" <<'PY' | tee -a "$O/table.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "g16_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = len(d) // 3
last = d[-n:]
print(f"diag {sys.argv[2]:>3s} K={sys.argv[3]:>2s}: " + "  ".join(f"{v:6.3f}" for v in last) + "  ms   " + rows[-1]["Kernel_Name"][:40])
PY
  done
done
