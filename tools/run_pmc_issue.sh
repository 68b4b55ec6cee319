#!/bin/bash
# Where does a SIMD's issue time go?  Two SQ passes over one bench step, summed per generator kernel template.
# usage: tools/run_pmc_issue.sh <tag> [build dir]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="$1"
if [ -n "${2:-}" ]; then export VSP_LIB_PATH="$R/build/$2/libvispeech_hip.so"; fi
G0="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
G1="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"
G2="SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
i=0
for G in "$G0" "$G1" "$G2"; do
  O="$R/gpurun_out/$TAG/g$i"; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --pmc $G --output-format csv -d "$O" -o p -- python3 "$R/bench.py" --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2> "$O/err.txt" || tail -3 "$O/err.txt"
  i=$((i+1))
done
python3 - "$R"/gpurun_out/$TAG/g*/p_counter_collection.csv <<'PY'
import csv, sys, collections, re
t = collections.OrderedDict()
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows}); half = ids[len(ids) // 2]
    for r in rows:
        if int(r["Dispatch_Id"]) < half: continue
        m = re.search(r"(g16_\w+<[^>]*>|conv1d_f32_mfma<[^>]*>)", r["Kernel_Name"])
        if not m: continue
        d = t.setdefault(m.group(1).replace(" ", ""), {})
        key = r["Counter_Name"]
        if key == "SQ_WAVE_CYCLES" and path != sys.argv[1]: continue
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
for k, d in t.items():
    cu = d["SQ_BUSY_CU_CYCLES"]; simd = 4 * cu
    g = lambda n: d.get(n, float("nan"))
    print(f"== {k}")
    print(f"   per SIMD-cycle: MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/simd:.3f} | VALU inst cycles {g('SQ_INST_CYCLES_VALU')/simd:.3f} (active-inst VALU {4*g('SQ_ACTIVE_INST_VALU')/simd:.3f}) | LDS active-inst {4*g('SQ_ACTIVE_INST_LDS')/simd:.3f} | VMEM inst cycles {g('SQ_INST_CYCLES_VMEM')/simd:.3f} (active {4*g('SQ_ACTIVE_INST_VMEM')/simd:.3f}) | scalar {4*g('SQ_ACTIVE_INST_SCA')/simd:.3f} | misc {4*g('SQ_ACTIVE_INST_MISC')/simd:.3f} | MFMA+VALU coexec {g('SQ_VALU_MFMA_COEXEC_CYCLES')/simd:.3f}")
    print(f"   instructions per MFMA: VALU {g('SQ_INSTS_VALU')/g('SQ_INSTS_MFMA') - 1:.2f}  SALU {g('SQ_INSTS_SALU')/g('SQ_INSTS_MFMA'):.2f}  LDS {g('SQ_INSTS_LDS')/g('SQ_INSTS_MFMA'):.2f}  SMEM {g('SQ_INSTS_SMEM')/g('SQ_INSTS_MFMA'):.3f}   (MFMA {g('SQ_INSTS_MFMA'):.3g}; SQ_INSTS_VALU counts the MFMAs too)")
    wc = d["SQ_WAVE_CYCLES"]
    print(f"   wave cycles: waiting {g('SQ_WAIT_ANY')/wc:.2f}  issue-stalled {g('SQ_WAIT_INST_ANY')/wc:.2f} (of which LDS {g('SQ_WAIT_INST_LDS')/wc:.2f})  issuing {g('SQ_ACTIVE_INST_ANY')/wc:.2f}   waves/CU {4*wc/cu:.1f}")
PY
