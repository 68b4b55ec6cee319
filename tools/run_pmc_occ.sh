#!/bin/bash
# occupancy / matrix-pipe duty of one library build: SQ counters over one bench step.  usage: tools/run_pmc_occ.sh <tag> [build dir]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="$1"
if [ -n "${2:-}" ]; then export VSP_LIB_PATH="$R/build/$2/libvispeech_hip.so"; fi
O="$R/gpurun_out/$TAG"; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O" -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2> "$O/err.txt" || tail -3 "$O/err.txt"
python3 - "$O/p_counter_collection.csv" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ids = sorted({int(r["Dispatch_Id"]) for r in rows}); half = ids[len(ids) // 2]
t = collections.OrderedDict()
for r in rows:
    if int(r["Dispatch_Id"]) < half: continue
    m = re.search(r"(g16_\w+<[^>]*>)", r["Kernel_Name"])
    if not m: continue
    d = t.setdefault(m.group(1).replace(" ", ""), collections.Counter())
    d[r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in t.items():
    cu = d["SQ_BUSY_CU_CYCLES"]
    print(f"{k:34s} waves/CU {4 * d['SQ_WAVE_CYCLES'] / cu:5.1f}  MFMA busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * cu):.3f}  wait {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.2f}  stall {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.2f}  issue {d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES']:.2f}  time {d['GRBM_GUI_ACTIVE'] / 8 / 1.9e6:6.2f} ms@1.9GHz")
PY
